for B in 4 8; do
VBX_SHARD_BANDS=$B timeout 300 python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('bands $B:', d['value'], d['ms_per_step'], d['exchange'])"
done
VBX_SHARD_BANDS=4 timeout 300 python bench.py --workload sensors4 --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --exchange native 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('bands 4 native:', d['value'], d['ms_per_step'], d['exchange'])"
