run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-host-path --profile-frames 0 --mirror-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stage_ms']['replay_ms'], d['counters_per_step']['replay_rounds'], d['counters_per_step']['replay_block_rounds'])"; }
echo default; run
echo no_blocks; VBX_REPLAY_NO_BLOCKS=1 run
for B in 4 8 32; do echo blocks $B; VBX_REPLAY_BLOCKS=$B run; done
echo batch4; VBX_REPLAY_BATCH=4 run
