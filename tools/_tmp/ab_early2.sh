run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-host-path --profile-frames 0 --mirror-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stage_ms']['replay_ms'], d['counters_per_step']['replay_rounds'], d['counters_per_step']['replay_block_rounds'])"; }
echo default; run
for M in 300000 800000; do for B in 2 4 8; do echo min $M blocks $B; VBX_REPLAY_BLOCKS_MIN=$M VBX_REPLAY_BLOCKS=$B run; done; done
