"""The host half under ThreadSanitizer (SURVEY 8(b) "Threading"): the product's own csrc/host/*.cpp
built with g++ -fsanitize=thread behind tests/sanitize/host_threads.cpp.  Four host threads each open their OWN rdoom_wad
handle (the reference's Archive is !Sync, wad/src/archive.rs:21: a handle is single-owner), build every level of the
synthetic IWAD several times and digest the arrays, one of them provoking errors in between; the main thread flips a debug
hook meanwhile.  TSan must stay silent, every thread must compute the same digest, and an error message must stay on the
thread that caused it (rdoom_last_error is thread-local)."""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from util import META_PATH, ROOT, ensure_wad

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, 'sanitize', '_build')
DRIVER = os.path.join(BUILD, 'host_threads_tsan')


@pytest.fixture(scope='module')
def tsan_driver():
    host = os.path.join(ROOT, 'rust-doom_amd', 'csrc', 'host')
    srcs = sorted(glob.glob(os.path.join(host, '*.cpp'))) + [os.path.join(HERE, 'sanitize', 'host_threads.cpp')]
    deps = srcs + glob.glob(os.path.join(host, '*.hpp')) + [os.path.join(ROOT, 'include', 'rdoom.h'),
                                                            os.path.join(ROOT, 'rust-doom_amd', 'csrc', 'common.hpp')]
    os.makedirs(BUILD, exist_ok=True)
    import fcntl
    lock = open(os.path.join(BUILD, '.lock_tsan'), 'w')  # (pytest-xdist: one worker builds, the others wait)
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if os.path.exists(DRIVER) and all(os.path.getmtime(d) <= os.path.getmtime(DRIVER) for d in deps):
            return DRIVER
        flags = ['-std=c++17', '-O1', '-g', '-fsanitize=thread', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'), '-I' + host,
                 '-I' + os.path.join(ROOT, 'rust-doom_amd', 'csrc')]

        def cc(src):
            obj = os.path.join(BUILD, 'tsan_' + os.path.basename(src) + '.o')
            subprocess.check_call(['g++'] + flags + ['-c', src, '-o', obj])
            return obj

        with ThreadPoolExecutor(min(len(srcs), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(cc, srcs))
        subprocess.check_call(['g++', '-fsanitize=thread', '-pthread'] + objs + ['-o', DRIVER])
        return DRIVER
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def test_host_library_from_four_threads_under_tsan(tsan_driver):
    env = dict(os.environ, TSAN_OPTIONS='halt_on_error=1 exitcode=66')
    p = subprocess.run([tsan_driver, ensure_wad(), META_PATH, '4', '3'], capture_output=True, text=True, timeout=600, env=env)
    assert 'ThreadSanitizer' not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    digests = re.findall(r'DIGEST (\d+) ([0-9a-f]{8})', p.stdout)
    assert len(digests) == 4 and len({d for _, d in digests}) == 1 and digests[0][1] != '00000000', p.stdout
    m = re.search(r'FAILURES (\d+) FLIPS (\d+)', p.stdout)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) > 0, p.stdout
