python tools/exp_phase.py 40 2>&1 | tail -12
