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


@pytest.mark.parametrize("src", ["conv_sc_lean.hip", "conv_sc.hip", "conv_pw.hip"])
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


def store_hazards(asm, slots=2):
    """128-bit buffer stores with a register soffset whose data registers a VALU instruction overwrites within ``slots`` issue slots
    (s_nop N counts N + 1): -> (stores seen, violations [(kernel, line, store, overwriting instruction)])."""
    kernel, watch, bad, nst = None, [], [], 0            # watch: [remaining slots, data registers, store text]
    for ln, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";", 1)[0].strip()
        if not line:
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):$", line)
        if m:
            if not m.group(1).startswith(".L"):
                kernel, watch = m.group(1), []
            continue
        if line.startswith(".") or kernel is None:
            continue
        op = line.split()[0]
        args = line[len(op):]
        if op.startswith("v_") and not op.startswith("v_cmp"):
            dest = vregs(args.split(",")[0])
            for w in watch:
                if dest & w[1]:
                    bad.append((kernel, ln, w[2], line))
        step = int(re.search(r"s_nop\s+(\d+)", line).group(1)) + 1 if op == "s_nop" else 1
        watch = [[w[0] - step, w[1], w[2]] for w in watch if w[0] - step > 0]
        if op == "buffer_store_dwordx4":
            parts = [a_.strip() for a_ in args.split(",")]
            if len(parts) >= 4 and re.match(r"^s\d+", parts[3].split()[0]):        # vdata, vaddr, srsrc, soffset(register) ...
                nst += 1
                watch.append([slots, vregs(parts[0]), line])
    return nst, bad


def test_wide_buffer_stores_keep_their_data_registers_for_two_slots(tmp_path):
    """Round 6, found on MI355X (conv_pw.hip): a buffer_store_dwordx4 with an SGPR soffset still reads its data registers in the slots after
    issue; a VALU write to the first data register in the next slot corrupted that dword in the last four lanes of every 16-lane row.
    hipcc inserts a wait state only for stores WITHOUT a register soffset, so the kernel pins two idle slots behind every such store -
    this test holds the generated code to it."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "conv_pw.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-Wno-unused-result", "--cuda-device-only", "-S",
                        os.path.join(CSRC, "conv_pw.hip"), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        nst, bad = store_hazards(f.read())
    assert nst > 100, nst
    assert not bad, "data registers of a wide buffer store overwritten within two slots:\n" + "\n".join("%s line %d: %s | %s" % b for b in bad[:20])


def test_the_store_scanner_sees_the_hazard():
    asm = """
kern:
	buffer_store_dwordx4 v[54:57], v40, s[12:15], s17 offen
	v_lshlrev_b32_e32 v54, 16, v42
	buffer_store_dwordx4 v[42:45], v41, s[12:15], s17 offen
	s_nop 1
	v_lshlrev_b32_e32 v42, 16, v42
	buffer_store_dwordx4 v[46:49], v41, s[12:15], 0 offen
	v_lshlrev_b32_e32 v46, 16, v42
	s_endpgm
"""
    nst, bad = store_hazards(asm)
    assert nst == 2 and [b[1] for b in bad] == [4]
