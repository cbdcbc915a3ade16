"""The C-ABI library loads and exports every symbol include/whisper_amd.h declares.
No compute call is made here: this runs without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from whisper_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    return sorted(set(re.findall(r"WAMD_API\s+[\w\s\*]+?\b(wamd_\w+)\s*\(", text)))


def test_header_symbols_exported():
    names = declared_symbols()
    assert len(names) >= 20
    L = api.lib()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(api.EXPORTS) == names


def test_struct_layouts_match_header():
    assert api.DIST_TASK.itemsize == 16
    assert api.DIST_RESULT.itemsize == 8
    assert api.LEVDIAG_TASK.itemsize == 16
    assert api.ALIGN_RESULT.itemsize == 40
    assert ctypes.sizeof(api.Scoring) == 56


def test_version_and_errors_without_gpu():
    assert "gfx950" in api.version()
    if api.device_count() == 0:
        # the product path fails loudly when there is no device: no CPU fallback
        with pytest.raises(api.WamdError) as e:
            api.Context(np.zeros(1024, np.uint8))
        assert e.value.status == 3
