"""Builds profiles/r02_* from one run of scripts/r02_evidence.sh (gpurun_out/r02_evidence): bench lines, rocprofv3 kernel stats and
one frame's timeline, per-kernel counter sums per frame (fabric requests, L2, writes, SQ), the primary kernel's VALU-busy figure
at 4K, rank shares.  usage: python scripts/make_r02_profiles.py [evidence dir]"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_evidence")
dst = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def kname(n):
    n = n.strip('"')
    if n.startswith("void "): n = n[5:]
    return n.split("(")[0].split("<")[0]


def counters(tag):
    """{kernel: {counter: sum}}, {kernel: dispatches} of one --pmc pass"""
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in glob.glob(os.path.join(src, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); calls[k] += 1
    return sums, calls


for a, b in [("bench_lambert1m", "r02_bench"), ("bench_lambert_4k", "r02_bench_lambert4k"), ("bench_lambert_10m_4k", "r02_bench_lambert10m_4k"),
             ("bench_s4", "r02_bench_s4"), ("bench_primary_100k", "r02_bench_primary100k")]:
    line = open(os.path.join(src, a + ".json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, b + ".json"), "w").write(line + "\n")

stats = glob.glob(os.path.join(src, "ktrace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, "r02_wavefront_lambert1m_kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "ktrace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace)) if kname(r["Kernel_Name"]).startswith(("wf_", "rt_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
start = idx[len(idx) // 2] # a frame of the timed loop (the last ones are the stage-timed frames: events between the stages)
t0 = int(rows[start]["Start_Timestamp"])
with open(os.path.join(dst, "r02_frame_timeline.txt"), "w") as f:
    f.write("# one planned frame of lambert_1m (rocprofv3 --kernel-trace of `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`): start offset, duration, kernel\n")
    for r in rows[start:idx[len(idx) // 2 + 1]]:
        f.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f}us  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f}us  {kname(r["Kernel_Name"])}\n')

per = collections.defaultdict(dict)
ea, calls = counters("ea")
frames = calls["wf_primary_kernel"]
hit, _ = counters("hit")
wr, _ = counters("write")
sq, _ = counters("sq")
for k in ea:
    if not k.startswith(("wf_", "rt_detile")): continue
    e, h, w, s = ea[k], hit.get(k, {}), wr.get(k, {}), sq.get(k, {})
    per[k] = {"fabric_read_requests": round(e["TCC_EA0_RDREQ_sum"] / frames), "fabric_read_bytes": round(e["TCC_EA0_RDREQ_sum"] * 128 / frames),
              "requests_128B": round(e["TCC_EA0_RDREQ_128B_sum"] / frames), "requests_64B": round(e["TCC_EA0_RDREQ_64B_sum"] / frames),
              "requests_32B": round(e["TCC_EA0_RDREQ_32B_sum"] / frames), "l2_requests": round(h.get("TCC_REQ_sum", 0) / frames),
              "l2_hits": round(h.get("TCC_HIT_sum", 0) / frames), "l2_misses": round(h.get("TCC_MISS_sum", 0) / frames),
              "write_bytes": round(w.get("WRITE_SIZE", 0) * 1024 / frames) if w.get("WRITE_SIZE", 0) < 1e9 else round(w.get("WRITE_SIZE", 0) / frames),
              "valu_wave_instructions": round(s.get("SQ_INSTS_VALU", 0) / frames), "valu_thread_cycles": round(s.get("SQ_THREAD_CYCLES_VALU", 0) / frames),
              "wave_cycles_x4": round(s.get("SQ_WAVE_CYCLES", 0) / frames), "wait_any_x4": round(s.get("SQ_WAIT_ANY", 0) / frames),
              "wait_inst_any_x4": round(s.get("SQ_WAIT_INST_ANY", 0) / frames)}
old = json.load(open(os.path.join(dst, "r02_trace_fabric_traffic.json")))
json.dump({"workload": "lambert_1m", "commit": commit, "frames": frames, "note": old["note"], "per_frame": per},
          open(os.path.join(dst, "r02_trace_fabric_traffic.json"), "w"), indent=1)

sq4, c4 = counters("sq4k")
g4, _ = counters("grbm4k")
f4 = c4["wf_primary_kernel"]
p = {k: round(v / f4) for k, v in sq4["wf_primary_kernel"].items()}
p.update({k: round(v / f4) for k, v in g4["wf_primary_kernel"].items()})
cyc = p["GRBM_GUI_ACTIVE"] / 8
valu = p["SQ_INSTS_VALU"] / 1024 * 4
reading = ("GRBM_GUI_ACTIVE sums the 8 XCDs: /8 = cycles the kernel took. SQ_INSTS_VALU / 1024 SIMDs x 4 cycles per wave64 instruction = cycles of "
           "VALU issue per SIMD. Their ratio is the share of the kernel's duration in which every SIMD was issuing vector instructions: at ~1.0 "
           "the kernel sits on its instruction-issue floor (an earlier reading added 12 cycles for each of a wave's 76 32-bit multiplies as "
           "quarter-rate instructions; with the kernel at this speed that sum would exceed its duration, so they cannot cost that much here).  "
           "The two counters come from separate profiler passes: a ratio a few percent either side of 1 is their run-to-run noise.")
json.dump({"workload": "lambert_4k", "commit": commit, "kernel": "wf_primary_kernel", "per_frame": p, "reading": reading,
           "derived": {"kernel_cycles": round(cyc), "valu_issue_cycles_per_simd": round(valu), "valu_busy_fraction": round(valu / cyc, 3),
                       "instructions_per_wave": round(p["SQ_INSTS_VALU"] / p["SQ_WAVES"], 1)}},
          open(os.path.join(dst, "r02_primary_4k_valu.json"), "w"), indent=1)

open(os.path.join(dst, "r02_rank_share.txt"), "w").write(open(os.path.join(src, "rank_share_1m.txt")).read() + open(os.path.join(src, "rank_share_4k.txt")).read())
if os.path.exists(os.path.join(src, "valu_busy.json")):
    v = json.load(open(os.path.join(src, "valu_busy.json")))
    v["commit"] = commit
    v["note"] = "per kernel and frame: GRBM_GUI_ACTIVE / 8 XCDs = cycles the kernel took; SQ_INSTS_VALU x 4 / 1024 SIMDs = cycles of VALU issue per SIMD (scripts/pmc_valu.sh)"
    json.dump(v, open(os.path.join(dst, "r02_valu_busy.json"), "w"), indent=1)
print("profiles/r02_* rebuilt from", src, "at", commit, "frames", frames)
