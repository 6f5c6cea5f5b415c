"""Loads oracle/liboracle.so (CPU restatement; TEST INFRASTRUCTURE ONLY) behind the same Backend API."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def open_oracle():
    from mvil_fusion_amd import lib
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    return lib.Backend(C.CDLL(ORACLE_SO), "orc_")
