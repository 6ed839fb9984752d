run() { VBX_FORCE_SHARDED=1 VBX_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['exchange']['integrate_ms_per_step'], d['exchange']['exchange_ms_per_step'])"; }
for i in 1 2; do echo keep; run; echo full; VBX_DELTA_FULL_CLEAR=1 run; done
echo plain; python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-host-path --profile-frames 0 --mirror-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
