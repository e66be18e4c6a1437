export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/x_kt$i.json 2> gpurun_out/x_kt$i.err
python3 -c "
import json;d=json.loads(open('gpurun_out/x_kt$i.json').read().strip().splitlines()[-1]);print('run $i',d['ms_per_step'],d['roofline']['frame']['device_ms'],d['roofline']['frac'],d['roofline']['stage_ms_per_frame'])"
done
