for Q in 4 8 16; do
GPU_MAX_HW_QUEUES=$Q python bench.py --workload sensors4 --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --profile-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('hw queues $Q:', d['value'], d['ms_per_step'])"
done
python bench.py --workload sensors4 --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --profile-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('default:', d['value'], d['ms_per_step'])"
