export TMPDIR=/tmp
mkdir -p gpurun_out/x_bench
for n in 2 4; do
RT_BENCH_REHEARSE=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/x_bench/n$n.json 2> gpurun_out/x_bench/n$n.err; echo "rehearsal N=$n rc $?"
python3 -c "
import json;d=json.loads(open('gpurun_out/x_bench/n$n.json').read().strip().splitlines()[-1]);print(d['n_gpus'],d['ms_per_step'],d['scaling'],d['parity'],d['config']['parallelism'])"
done
