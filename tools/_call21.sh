cp rust-doom_amd/librdoom_hip.so /tmp/_s.so; cp _variants/fstats.so rust-doom_amd/librdoom_hip.so
python bench.py --streams 1 --steps 1 --warmup 0 --poses 256 --cpu-sample 0 --other off 2>&1 | grep -a "frag stats" | tail -3
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
