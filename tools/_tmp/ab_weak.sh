run() { VBX_FORCE_SHARDED=1 VBX_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['exchange']['integrate_ms_per_step'], d['exchange']['exchange_ms_per_step'])"; }
for i in 1 2; do echo full; run; echo keep; VBX_DELTA_KEEP=1 run; done
