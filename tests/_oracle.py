"""Test helper: the CPU oracle exposed through the same ctypes ABI wrapper.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

from __future__ import annotations

import os
import subprocess

import numpy as np

from pymc_bart_amd import _abi
from pymc_bart_amd.sampler import Backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libpgbart_oracle.so")


class NumpyMemory:
    """"Device" memory of the CPU oracle: plain numpy arrays."""

    stream_ptr = None

    @staticmethod
    def from_host(arr):
        return np.array(arr, copy=True, order="C")

    @staticmethod
    def empty(shape, dtype=np.float64):
        return np.zeros(shape, dtype=dtype)

    @staticmethod
    def host_result(n):
        return np.empty(int(n), np.float64)

    @staticmethod
    def ptr(buf):
        return buf.ctypes.data

    @staticmethod
    def to_host(buf):
        return np.array(buf, copy=True)

    @staticmethod
    def synchronize():
        pass


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "pgbart_oracle.c")
    stale = (not os.path.exists(ORACLE_SO)) or any(
        os.path.getmtime(f) > os.path.getmtime(ORACLE_SO)
        for f in [src] + [os.path.join(ROOT, "include", h) for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                          if h.endswith(".h")]
    )
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B"])
    return ORACLE_SO


def build_oracle_native() -> tuple[str, str]:
    """The oracle compiled for THIS host's cores (bench.py's cpu_baseline leg: the stated baseline must
    not be a strawman).  -ffp-contract=off stays: the numeric contract forbids fusing, so the native
    build produces the same bits.  Falls back to the portable build if the compiler refuses."""
    src = os.path.join(ORACLE_DIR, "pgbart_oracle.c")
    out = os.path.join(ORACLE_DIR, "libpgbart_oracle_native.so")
    flags = "gcc -O3 -march=native -std=gnu11 -ffp-contract=off"
    try:
        subprocess.check_call(flags.split() + ["-fPIC", "-shared", "-DPGB_MAX_PARTICLES=128", "-I" + os.path.join(ROOT, "include"), src,
                                               "-o", out, "-lm"], stderr=subprocess.DEVNULL)
        return out, flags
    except (subprocess.CalledProcessError, OSError):
        return build_oracle(), "gcc -O2 -std=gnu11 -ffp-contract=off (portable build; -march=native failed)"


_BACKEND = None


def oracle_backend() -> Backend:
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = Backend(lib=_abi.PGBLibrary(build_oracle()), mem=NumpyMemory())
    return _BACKEND
