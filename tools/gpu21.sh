timeout 1500 python -m pytest tests/test_golden_wtns.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
