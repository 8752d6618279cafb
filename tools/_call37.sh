set -u
bash tools/stress_round.sh r04v 2>&1 | tail -3
bash tools/profile_round.sh r04v 2>&1 | tail -4
