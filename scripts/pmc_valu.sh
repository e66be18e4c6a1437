# usage: bash scripts/pmc_valu.sh <tag> [workload] -- one counter pass of bench.py: cycles each kernel took (GRBM_GUI_ACTIVE / 8 XCDs)
# against its VALU issue cycles (SQ_INSTS_VALU x 4 / 1024 SIMDs); prints the VALU-busy fraction per wf_ kernel and frame
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcv_$tag
mkdir -p $out
timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/run -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1 || echo "counter pass failed"
python3 - $out <<'PY'
import csv, glob, sys, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/run/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (k, r["Dispatch_Id"])
        if d not in seen: seen.add(d); calls[k] += 1
frames = calls.get("wf_primary_kernel", 1)
out = {}
for k, c in rows.items():
    if not k.startswith("wf_"): continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8 / frames
    valu = c["SQ_INSTS_VALU"] * 4 / 1024 / frames
    out[k] = {"kernel_cycles": round(cyc), "valu_issue_cycles_per_simd": round(valu), "valu_busy": round(valu / cyc, 3) if cyc else None,
              "insts_valu": round(c["SQ_INSTS_VALU"] / frames), "insts_salu": round(c["SQ_INSTS_SALU"] / frames),
              "insts_lds": round(c["SQ_INSTS_LDS"] / frames), "insts_vmem_rd": round(c["SQ_INSTS_VMEM_RD"] / frames), "launches_per_frame": calls[k] / frames}
    print(k, json.dumps(out[k]))
json.dump({"frames": frames, "per_frame": out}, open(sys.argv[1] + "/valu.json", "w"), indent=1)
PY
