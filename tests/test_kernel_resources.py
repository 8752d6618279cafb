"""The shipped kernels' register / scratch footprint, read from the code objects inside rust-doom_amd/librdoom_hip.so
(tools/kernel_resources.py: llvm-objdump --offloading + llvm-readelf --notes).  The hot kernels may not touch scratch
memory: a spilled VGPR in the rasteriser's hot instantiation was a finding of round 2's review (it came from run-time flags
the compiler kept as per-lane booleans; they are template parameters now)."""
import importlib.util
import os
import shutil

import pytest

from util import ROOT

_spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(ROOT, 'tools', 'kernel_resources.py'))
kr = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(kr)


@pytest.mark.skipif(not (os.path.exists(os.path.join(kr.LLVM, 'llvm-objdump')) and shutil.which('c++filt')), reason='needs the ROCm LLVM tools')
def test_hot_kernels_use_no_scratch():
    res = {kr.short(k): v for k, v in kr.kernel_resources().items()}
    hot = [k for k in res if k.startswith(('raster_wave_kernel<false', 'fragment_kernel<', 'setup_kernel', 'cull_kernel', 'bin_kernel<',
                                           'sort_scan_kernel', 'fixup_kernel'))]
    assert len(hot) >= 18, sorted(res)
    for k in hot:
        r = res[k]
        assert r['private_segment_fixed_size'] == 0 and r['vgpr_spill_count'] == 0, (k, r)
    # occupancy targets the launch bounds state: 4 waves per SIMD for the rasteriser (128 VGPRs), 6 for the fragment kernel (80)
    assert all(res[k]['vgpr_count'] <= 128 for k in res if k.startswith('raster_wave_kernel<false'))
    assert all(res[k]['vgpr_count'] <= 80 for k in res if k.startswith('fragment_kernel<'))
    # LDS per workgroup: COLORMAP (8 KiB) + the per-wave quad lists in the fragment kernel; the parked records in the rasteriser
    assert res['fragment_kernel<2, 0, true>']['group_segment_fixed_size'] <= 12 * 1024
    for split in ('false', 'true'):  # (the instantiation without / with the per-quadrant lists of long tiles)
        assert res['raster_wave_kernel<false, true, false, true, %s>' % split]['group_segment_fixed_size'] <= 6 * 1024  # (no stats, 16-bit words, no ids, SKIPVIS;
    # 4 KiB of parked records + 1 KiB of edge hashes / cover-but-for-one-edge bits + the list scratch: 16 one-wave workgroups per CU use 84 of 160 KiB)
