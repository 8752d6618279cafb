"""Pins taken from the reference's own tests.  The reference holds exactly one known-answer test that touches
this path: `test_wad_name` (wad/src/name.rs:163-190).  Its 15 asserts are reproduced verbatim here and run
against BOTH the numpy oracle (oracle/wad_oracle.py:wad_name) and the product's C++ implementation behind the
C ABI (rdoom_wad_name_from_bytes).  Everything else on the path is unpinned by the reference."""
import pytest

import rust_doom_amd as rd
from oracle import wad_oracle

# (input, expected 8 bytes)  -- wad/src/name.rs:168-182
OK = [
    (b'', b'\0\0\0\0\0\0\0\0'),
    (b'\0', b'\0\0\0\0\0\0\0\0'),
    (b'\x001234567', b'\0\0\0\0\0\0\0\0'),
    (b'A', b'A\0\0\0\0\0\0\0'),
    (b'1234567', b'1234567\0'),
    (b'12345678', b'12345678'),
    (b'123\x005678', b'123\0\0\0\0\0'),
    (b'SKY1', b'SKY1\0\0\0\0'),
    (b'-', b'-\0\0\0\0\0\0\0'),
    (b'_', b'_\0\0\0\0\0\0\0'),
]
# wad/src/name.rs:184-188
ERR = [b'123456789', b'1234\xfb', b'\xff123', b'$$ASDF_', b'123456789\0']


@pytest.mark.parametrize('value,want', OK)
def test_wad_name_ok(value, want):
    assert wad_oracle.wad_name(value) == want
    assert rd.wad_name(value) == want


@pytest.mark.parametrize('value', ERR)
def test_wad_name_err(value):
    with pytest.raises(wad_oracle.WadError):
        wad_oracle.wad_name(value)
    with pytest.raises(rd.RdoomError):
        rd.wad_name(value)


def test_wad_name_uppercases():
    """from_bytes upper-cases ASCII letters (name.rs:50) -- not in the reference test, follows from the source."""
    assert wad_oracle.wad_name(b'sky1') == b'SKY1\0\0\0\0' == rd.wad_name(b'sky1')
    assert wad_oracle.wad_name(b'f_sky1') == b'F_SKY1\0\0' == rd.wad_name(b'f_sky1')
