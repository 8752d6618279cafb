set -u
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/prof_sizes.sh r04u 2>&1 | grep -v "^$" | cut -c1-160
