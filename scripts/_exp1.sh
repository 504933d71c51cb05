B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
show() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s value %.0f first %.0f' % (sys.argv[1], d['value'], d.get('first_window',{}).get('value',0)))" "$1"; }
$B 2>/dev/null | tail -1 | show default_runtime_chunk3
SO101_NARROW_CHUNK=4 $B 2>/dev/null | tail -1 | show chunk4
SO101_NARROW_CHUNK=2 $B 2>/dev/null | tail -1 | show chunk2
$B --envs-per-gpu 16384 2>/dev/null | tail -1 | show default_16384
SO101_NARROW_CHUNK=3 $B --envs-per-gpu 16384 2>/dev/null | tail -1 | show chunk3_16384
SO101_NARROW_CHUNK=3 $B --envs-per-gpu 32768 2>/dev/null | tail -1 | show chunk3_32768
$B --envs-per-gpu 32768 2>/dev/null | tail -1 | show default_32768
