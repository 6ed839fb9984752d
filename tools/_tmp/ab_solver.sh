O=gpurun_out/r03y; mkdir -p $O
for K in 3 5 8; do
  VBX_SOLVER_SWEEPS=$K timeout 200 python bench.py --no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 > $O/b_$K.json 2> $O/b_$K.err
  python -c "
import json;d=json.loads(open('$O/b_$K.json').read().strip().splitlines()[-1]);print('sweeps $K:',d['value'],d['ms_per_step'],d.get('stage_ms'),d.get('counters_per_step'))"
done
VBX_SOLVER_SWEEPS=5 VBX_SOLVER_OPEN_GUESS=l timeout 200 python bench.py --no-cpu-baseline --no-extras --no-host-path --mirror-frames 0 > $O/b_5l.json 2> $O/b_5l.err
python -c "
import json;d=json.loads(open('$O/b_5l.json').read().strip().splitlines()[-1]);print('sweeps 5 tl:',d['value'],d['ms_per_step'],d.get('stage_ms'))"
for K in 4 8; do
VBX_SOLVER_SWEEPS=$K timeout 300 python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > $O/s4_$K.json 2> $O/s4_$K.err
python -c "
import json;d=json.loads(open('$O/s4_$K.json').read().strip().splitlines()[-1]);print('s4 sweeps $K:',d['value'],d['ms_per_step'])"
done
