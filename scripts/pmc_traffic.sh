# usage: bash scripts/pmc_traffic.sh <tag> [workload] -- HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (TCC slots), no tracing domains mixed in; writes gpurun_out/traffic_<tag>/traffic.json
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/traffic_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - $out $wl <<'PY'
import csv, glob, json, sys, collections
root, wl = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not (k.startswith("wf_") or k.startswith("rt_")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": calls[k] += 1
frames = calls.get("wf_primary_kernel", 0) or 1
res = {"workload": wl, "frames": frames, "note": "KB counters summed over all dispatches of the run, divided by frames; "
       "gfx950: FETCH_SIZE tallies 128-B fabric reads at 64 B, so read bytes = 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM)", "per_frame": {}}
for k, c in tot.items():
    rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024 / frames; wr = c.get("WRITE_SIZE", 0.0) * 1024 / frames
    res["per_frame"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "dispatches_per_frame": calls[k] / frames}
json.dump(res, open(root + "/traffic.json", "w"), indent=1)
for k, v in sorted(res["per_frame"].items()): print(f'{k:32s} {v["hbm_bytes"]/1e6:10.1f} MB/frame  (read {v["read_bytes"]/1e6:.1f}, write {v["write_bytes"]/1e6:.1f}; {v["dispatches_per_frame"]:.1f} dispatches)')
PY
