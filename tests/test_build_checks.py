"""Build-time checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

K4 (tod_amd/csrc/match.hip, hamming_topk_tiles) issues its DB-row prefetches as hand-written `s_load_dwordx16` pairs and waits
for them with a hand-written `s_waitcnt lgkmcnt(0)`; hipcc does not know that the destination SGPRs are in flight in
between. The design is only correct while the compiler neither reads, copies nor spills those registers between an issue
and its wait -- which nothing but the generated code can confirm, so this test disassembles it (same compiler, same flags
as the Makefile) and checks every instruction between each s_load_dwordx16 and the following lgkmcnt(0) wait."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _sgprs(text):
    """SGPR numbers an instruction's operand text touches."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        regs.add(int(a))
    return regs


@pytest.fixture(scope="module")
def match_asm():
    """match.hip as gfx950 assembly, compiled once for the module (same compiler, same flags as the Makefile)"""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "match.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-o", out, os.path.join(ROOT, "tod_amd", "csrc", "match.hip")], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def test_k4x_waves_leave_room_in_the_register_file(match_asm):
    """The matrix-core matcher's default launch (six query blocks per wave, k <= 2) must stay at <= 224 registers per wave: two
    waves then leave >= 64 of a SIMD's 512 free and ORB's and the verifier's kernels start beside them instead of waiting for a
    matcher workgroup to retire (ORB's stage 1.2 ms against 1.8-1.9). A harmless-looking change of the hot loop once took
    the kernel to 243 registers and the pipeline lost that silently -- only the generated code can tell. No spills either."""
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S*hamming_topk_mfmaILi([12])ELi6ELi([0-3])ELb\d\S*)", match_asm):
        blk = match_asm[m.start():m.start() + 1500]
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert vgpr <= 224 and scratch == 0, "hamming_topk_mfma<%s, 6, %s>: %d registers, %d bytes of scratch" % (m.group(2), m.group(3), vgpr, scratch)
        seen += 1
    assert seen == 8, "expected the four block forms of hamming_topk_mfma<1 / 2, 6, ...>"


def test_k4_prefetched_sgprs_are_untouched_until_their_wait(match_asm):
    asm = match_asm
    kernels, cur = [], None                                            # (name, [instruction lines]) per instantiation
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN\S*hamming_topk_tiles\S*):", line)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur[1].append(line)
    assert len(kernels) >= 8 * 4, "expected every (K, MODE) instantiation of hamming_topk_tiles"
    n_windows = 0
    for name, lines in kernels:
        in_flight = set()
        for line in lines:
            ins = line.split(";")[0].strip()
            if not ins or ins.endswith(":") or ins.startswith("."):
                continue
            m = re.match(r"s_load_dwordx16 s\[(\d+):(\d+)\], (.*)", ins)
            if m:
                assert not (_sgprs(m.group(3)) & in_flight), "%s: address of `%s` lives in registers still in flight" % (name, ins)
                in_flight |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                n_windows += 1
                continue
            if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                in_flight.clear()
                continue
            if in_flight:
                touched = _sgprs(ins) & in_flight
                assert not touched, "%s: `%s` touches s%s while its s_load_dwordx16 is in flight" % (name, ins, sorted(touched))
    assert n_windows >= 2 * len(kernels)
