"""Static check of the device code (CPU, no GPU): no instruction may touch the destination registers of an LDS read before the s_waitcnt
that covers it.  The compiler guarantees this for its own reads; the small-channel kernels issue theirs from inline asm (a ring of
ds_read_b128 kept in flight ahead of the MFMAs, released one by one with counted waits: conv_sc.hip, conv_sc_lean.hip) and declare the ring
registers as plain outputs - nothing in the language stops a future compiler from copying or spilling such a register between the issue
and its wait (advisor finding, round 4).  This test compiles the two files to gfx950 assembly and walks every kernel with the LGKM
counter's in-order model (LDS operations return in order; lgkmcnt(N) = at most N outstanding)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "segmentation_training_pipeline_amd", "csrc")
LGKM = re.compile(r"^(ds_|s_load_|s_buffer_load_|s_memtime|s_memrealtime)")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(asm):
    """-> (kernels seen, LDS reads tracked, violations [(kernel, line number, instruction, registers)])."""
    kernel, fifo, bad, nk, nreads = None, [], [], 0, 0
    for ln, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";", 1)[0].strip()
        if not line:
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):$", line)
        if m:
            if not m.group(1).startswith(".L"):          # a function label: a new kernel starts
                kernel, fifo = m.group(1), []
                nk += 1
            continue
        if line.startswith(".") or kernel is None:
            continue
        op = line.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
            continue
        used = vregs(line[len(op):])
        pending = set().union(*fifo) if fifo else set()
        if used & pending:
            bad.append((kernel, ln, line, sorted(used & pending)))
        if LGKM.match(op):
            dest = set()
            if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")):
                dest = vregs(line[len(op):].split(",")[0])
                nreads += 1
            fifo.append(dest)
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
            fifo = []
    return nk, nreads, bad


@pytest.mark.parametrize("src", ["conv_sc_lean.hip", "conv_sc.hip"])
def test_async_lds_reads_are_not_touched_before_their_wait(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-Wno-unused-result", "--cuda-device-only", "-S",
                        os.path.join(CSRC, src), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        nk, nreads, bad = scan(f.read())
    assert nk >= 4 and nreads > 200, (nk, nreads)              # the walk saw the kernels and their LDS reads
    assert not bad, "registers of an LDS read in flight are touched before its wait:\n" + "\n".join(
        "%s line %d: %s (v%s)" % (k, ln, ins, regs) for k, ln, ins, regs in bad[:20])


def test_the_scanner_sees_a_hazard():
    asm = """
kern:
	ds_read_b128 v[4:7], v1 offset:16
	ds_read_b128 v[8:11], v1 offset:32
	s_waitcnt lgkmcnt(1)
	v_mov_b32_e32 v20, v5
	v_add_f32_e32 v21, v9, v20
	s_waitcnt lgkmcnt(0)
	v_add_f32_e32 v22, v9, v20
	s_endpgm
"""
    nk, nreads, bad = scan(asm)
    assert nk == 1 and nreads == 2 and [b[1] for b in bad] == [7] and bad[0][3] == [9]
