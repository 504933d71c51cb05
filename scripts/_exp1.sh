B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
show() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s value %.0f first %.0f' % (sys.argv[1], d['value'], d.get('first_window',{}).get('value',0)))" "$1"; }
$B 2>/dev/null | tail -1 | show default_h2_l4
$B --envs-per-gpu 32768 2>/dev/null | tail -1 | show 32768_h4_l4
SO101_NARROW_CHUNK=2 $B --envs-per-gpu 32768 2>/dev/null | tail -1 | show 32768_h2_l4
$B --envs-per-gpu 16384 2>/dev/null | tail -1 | show 16384_h4_l4
SO101_NARROW_CHUNK=2 $B --envs-per-gpu 16384 2>/dev/null | tail -1 | show 16384_h2_l4
SO101_NARROW_WAVES_Q=6 $B 2>/dev/null | tail -1 | show default_wavesq6
SO101_NARROW_WAVES_Q=10 $B 2>/dev/null | tail -1 | show default_wavesq10
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipelined or chains or single_env_with_more" 2>&1 | tail -2
