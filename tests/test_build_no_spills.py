"""Build hygiene of the iteration's kernels: none of them may spill vector registers.  A spill in k_step / k_sweep once cost 12 - 20 bytes of scratch per lane and
a measurable share of the iteration (the chain workgroup's table entries decoded into a second set of registers); 1.3 kB of scratch per lane (the parameter block
copied by value) cost 35 us per LAUNCH in round 2.  Read from the code object's metadata -- no GPU needed."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mvil-fusion_amd", "csrc", "libvilsolve.so")


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf")) and os.path.exists(LIB)),
                    reason="needs the ROCm LLVM tools and the built library")
def test_iteration_kernels_do_not_spill_vector_registers():
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so"); shutil.copy(LIB, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=td, check=True, capture_output=True)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "gfx950" in f]
        assert cos, "no gfx950 code object in the library"
        seen = {}
        for co in cos:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split(".name:")[1:]:
                name = blk.split()[0]
                m = re.search(r"\.vgpr_spill_count:\s*(\d+)", blk); p = re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk); v = re.search(r"\.vgpr_count:\s*(\d+)", blk)
                if m and p and v: seen[name] = (int(m.group(1)), int(p.group(1)), int(v.group(1)))
        hot = {n: s for n, s in seen.items() if re.match(r"_Z\d+k_(iter|step|sweep|reduce)", n)}
        assert any("k_iter" in n for n in hot) and any("k_step" in n for n in hot) and any("k_sweep" in n for n in hot), sorted(seen)
        for n, (spill, scratch, vgprs) in hot.items():
            assert spill == 0, "%s spills %d vector registers (%d B of scratch per lane)" % (n, spill, scratch)
            assert scratch <= 64, "%s needs %d B of scratch per lane" % (n, scratch)       # (k_iter: 36 B reserved for scalar-register spill slots that end up in vector lanes -- no scratch instruction in the kernel)
            assert vgprs <= 256, (n, vgprs)                                                    # two waves per SIMD: the 512-thread workgroups need it
