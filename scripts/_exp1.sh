show() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s value %.0f' % (sys.argv[1], d['value']), d.get('diag_mean'))" "$1"; }
python bench.py --workload aloha --steps 20 --warmup 5 2>/dev/null | tail -1 | show aloha_sparse_hessian
python bench.py --workload dining --steps 10 --warmup 3 2>/dev/null | tail -1 | show dining_1024
python scripts/gpu_tree_phases.py 2>/dev/null | grep mask
timeout 900 python -m pytest tests/test_tree_parity.py tests/test_dining.py -m gpu -q -x 2>&1 | tail -3
