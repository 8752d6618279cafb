set -u
bash tools/ab_cycles.sh abc2 fr0 new 2>&1 | grep "fragment_kernel\|variant\|round"
