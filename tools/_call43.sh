set -u
bash tools/ab_so.sh _variants/cur.so _variants/fr_ilp.so _variants/fr_mem.so _variants/ra_ilp.so _variants/ra_mem.so -- --other off --streams 1 2>&1
