#!/bin/bash
# Timing experiments on fragment_kernel (wrong images by design; _variants/texp.so = -DRDOOM_TIMING_EXPERIMENTS): what does the kernel
# stop paying for when its texel loads (2), framebuffer stores (4), COLORMAP look-ups (8) are compiled out?  Fragment stage by hipEvents.
# build the variant first:  bash tools/variant.sh texp fragment "-DRDOOM_TIMING_EXPERIMENTS"   (RDOOM_FRAG_DBG bits: 2, 4, 8 as above; 16 = the block stored as one contiguous run)
cp rust-doom_amd/librdoom_hip.so /tmp/_s.so; cp _variants/texp.so rust-doom_amd/librdoom_hip.so
for round in 1 2; do
for D in 0 2 4 8 6 10 14 16; do
  RDOOM_FRAG_DBG=$D python bench.py --streams 1 --steps 10 --warmup 2 --cpu-sample 0 --other off 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('r$round RDOOM_FRAG_DBG=$D fragment stage %.3f ms' % d['config']['kernels_ms']['fragment'])"
done; done
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
