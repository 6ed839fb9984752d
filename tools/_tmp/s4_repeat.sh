cat /proc/loadavg; nproc
for i in 1 2 3; do
python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --profile-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('native run $i:', d['value'], d['ms_per_step'])"
cat /proc/loadavg
done
python bench.py --workload sensors4 --gpus 1 --steps 12 --warmup 2 --no-cpu-baseline --profile-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('native 12 steps:', d['value'], d['ms_per_step'])"
