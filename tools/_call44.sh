set -u
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/stress_round.sh r04w 2>&1 | tail -3
bash tools/profile_round.sh r04w 2>&1 | tail -4
