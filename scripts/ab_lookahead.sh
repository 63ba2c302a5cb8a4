for i in 1 2; do
for la in 3 1 0; do
VFN_LOOKAHEAD=$la python bench.py --steps 99 --warmup 3 --no-cpu-baseline --min-timed-s 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LA=$la', d['value'], d['ms_per_step'], d['frame_ms'])"
done; done
