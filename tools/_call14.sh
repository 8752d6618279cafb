time bash tools/stress_round.sh r04
