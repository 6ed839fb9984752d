s4() { python bench.py --gpus 1 --workload sensors4 --steps 6 --warmup 2 --no-cpu-baseline --profile-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sensors4', d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo keep; s4; echo full; VBX_DELTA_FULL_CLEAR=1 s4; done
