"""Shared helpers for the -m gpu parity tests: one libdg16 context per session."""

import numpy as np
import pytest

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        import dg16_amd
        _ctx = dg16_amd.Context(0)
    return _ctx
