"""Invariants of the compiled kernels that the source relies on but the language cannot express (checked on the ISA hipcc emits, CPU only).

k_attn_resident64 prefetches the next task's q fragments with inline-asm loads the compiler does not count and waits for them with ONE
explicit vmcnt(4) at the top of a task (lrx_attn.hip).  That is only sound while hipcc keeps the fragments in the registers the loads
wrote: a register copy between the load and the wait would copy data that has not arrived.  The same kernel and the tiled one must stay
free of scratch (a scratch reload comes with a vmcnt(0) that serialises the output stores)."""
import os, re, shutil, subprocess, tempfile
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "lightretriever_amd", "csrc", "lrx_attn.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def attn_isa():
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_body(isa, mangled_prefix):
    m = re.search(r"^(%s\w*):[^\n]*\n(.*?)s_endpgm" % mangled_prefix, isa, re.S | re.M)
    assert m, mangled_prefix
    return m.group(2)


def regs_of(rng):
    a, b = rng.split(":")
    return set(range(int(a), int(b) + 1))


def test_resident_attention_keeps_prefetched_q_in_place(attn_isa):
    body = kernel_body(attn_isa, "_Z17k_attn_resident64")
    assert "scratch_" not in body
    dests = re.findall(r"global_load_dwordx4 v\[(\d+:\d+)\]", body)
    assert len(dests) >= 8 and len(set(dests)) == 4, dests          # the counted first load + the asm prefetch sites: the same four quads
    qregs = set().union(*(regs_of(r) for r in set(dests)))
    assert len(qregs) == 16
    for line in body.splitlines():
        line = line.split(";")[0]
        used = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
            used |= set(range(int(a), int(b) + 1))
        used |= {int(x) for x in re.findall(r"\bv(\d+)\b", line)}
        if used & qregs:
            assert "global_load_dwordx4" in line or "v_mfma" in line, line.strip()   # nothing copies, spills or recomputes them
    assert len(re.findall(r"s_waitcnt vmcnt\(4\)", body)) == 1
    assert len(re.findall(r"buffer_store_dwordx4", body)) == 4       # always-issued stores: the count the vmcnt(4) stands on


def test_tiled_attention_at_head_dim_128_has_no_scratch(attn_isa):
    for grp in (1, 2, 3, 4):
        assert "scratch_" not in kernel_body(attn_isa, "_Z20k_attn_varlen_causalILi128ELi%dE" % grp)


def test_work_list_attention_has_no_scratch_at_the_shapes_the_encoders_use(attn_isa):
    # (a spilled value reloaded inside the item loop waits, with its vmcnt(0), for every tile request in flight: DESIGN.md 5.2)
    for d, grps in ((128, (1, 2, 3, 4)), (64, (1, 2, 3, 4, 5, 6, 7, 8))):
        for grp in grps:
            assert "scratch_" not in kernel_body(attn_isa, "_Z13k_attn_streamILi%dELi%dE" % (d, grp)), (d, grp)
