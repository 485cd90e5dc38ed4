"""Properties of the generated gfx950 code that the kernels' inline assembly relies on, checked on the disassembly of
libs2r.so (no GPU needed: hipcc cross-compiles here)."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _code_objects():
    spec = importlib.util.spec_from_file_location("code_objects", os.path.join(ROOT, "tools", "code_objects.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def disassembly():
    import synth2_amd as s2
    s2.load_library()                      # builds libs2r.so if needed
    if not os.path.exists(_code_objects().OBJDUMP):
        pytest.skip("llvm-objdump not available")
    from synth2_amd import build as _b
    texts = _code_objects().disassemble(_b.LIB)
    assert len(texts) >= 10, "expected one code object per translation unit"
    return texts


def test_m0_is_ours_between_the_tile_stores(disassembly):
    """chunk_fast puts the wave's tile address into M0 once per chunk (tile_set_base) and then issues sixteen
    ds_write_addtid_b32 (address = M0 + offset + 4 * lane) from separate asm statements: nothing the compiler emits may
    write M0 — every M0 write in the library must be our `s_mov_b32 m0, sN` + `s_nop 0`, and nothing else may read it."""
    from synth2_amd import build as _b
    n_set, n_store = _b.check_m0_contract()          # (what every build of the library runs on itself: synth2_amd/build.py)
    assert n_set > 0 and n_store == 16 * n_set, (n_set, n_store)
    # ... and the check does see a violation: code that touches M0 anywhere else, or sets it without the wait state, is refused
    k = next(i for i, t in enumerate(disassembly) if re.search(r"s_mov_b32 m0, s\d+", t))
    lines = disassembly[k].splitlines()
    at = next(i for i, l in enumerate(lines) if re.search(r"s_mov_b32 m0, s\d+", l))
    assert "s_nop" in lines[at + 1]
    elsewhere = lines[:at] + ["\ts_mov_b32 m0, exec_lo"] + lines[at:]
    no_wait = lines[:at + 1] + ["\tv_mov_b32_e32 v0, v0"] + lines[at + 2:]
    for bad in (elsewhere, no_wait):
        with pytest.raises(RuntimeError):
            _b.check_m0_contract(texts=disassembly[:k] + ["\n".join(bad)] + disassembly[k + 1:])


def test_hot_chunk_has_no_compare_or_select(disassembly):
    """The branch-free chunk of the saw one-pole kernel is straight-line packed arithmetic: no v_cmp / v_cndmask (8
    cycles each for a lone wave, tools/ubench/issue_rates3.hip) inside the basic block of its steady-state variant
    (flat envelopes, offsets below 2^24, aligned chunks), except the select of the run's dead-lane mask."""
    best = 1 << 30
    smallest = 1 << 30
    seen = 0
    for text in disassembly:
        if "s2r_render_kernelILi1ELb0ELi0E" not in text:
            continue
        body = text[text.index("s2r_render_kernelILi1ELb0ELi0E"):]
        # split into basic blocks at branch targets / branches
        blocks, cur = [], []
        for l in body.splitlines():
            ins = l.split("//")[0].strip()
            if re.match(r"^[0-9a-f]+ <", ins):
                if cur:
                    blocks.append(cur)
                cur = []
                if "s2r_render_kernelILi1ELb0ELi0E" not in ins and not ins.endswith(">:") :
                    break
                continue
            if ins:
                cur.append(ins)
                if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
                    blocks.append(cur)
                    cur = []
        for b in blocks:
            if sum("ds_write_addtid_b32" in i for i in b) == 16:
                seen += 1
                best = min(best, sum(i.startswith("v_cmp") or i.startswith("v_cndmask") for i in b))
                smallest = min(smallest, len(b))
    # (the variants for offsets beyond 2^24 convert and compare per frame; the steady-state ones must not)
    assert seen >= 4
    # (objdump shows no labels: a "block" here may start with the loop's preheader, which selects the run's dead-lane
    # constants once — a handful of v_cndmask — and restores spilled scalars)
    assert best <= 6, best
    assert smallest <= 215, "the steady-state chunk grew to %d instructions (r3: ~170 for 16 frames + its loop preheader)" % smallest


def test_render_kernels_keep_their_state_in_registers():
    """The render kernels the default patch runs — saw + one-pole, no mod-to-pitch, 256-voice workgroups; whole fills,
    per-voice rows and timed events — have no private segment (scratch) and spill no vector register: the envelope stage
    records stay in VGPRs (DESIGN.md §2 — round 2's timed-event kernel had 88 bytes per lane, 3.4 x the algorithmic HBM
    traffic).  Scalars may spill into VGPR lanes (v_writelane / v_readlane, no memory): bounded here so that growth shows.
    (The 1 024-voice-workgroup variants — 128 registers per lane, only on request through s2r_config.block_voices — and
    the mod-to-pitch variants do have scratch: DESIGN.md §8.)"""
    import synth2_amd as s2
    s2.load_library()
    co = _code_objects()
    if not os.path.exists(co.READELF):
        pytest.skip("llvm-readelf not available")
    from synth2_amd import build as _b
    md = co.kernel_metadata(_b.LIB)
    onepole = {k: v for k, v in md.items() if "s2r_render_kernelILi1ELb0E" in k and "ELi256EEE" in k}
    assert len(onepole) >= 3, sorted(md)
    for name, v in onepole.items():
        assert v[".private_segment_fixed_size"] == 0, (name, v)
        assert v[".vgpr_spill_count"] == 0, (name, v)
        assert v[".vgpr_count"] <= 256, (name, v)
        assert v[".sgpr_spill_count"] <= 300, (name, v)
    # round 4: the pool-resident kernel of the same patch (one workgroup per compute unit: the whole register file), and the
    # general kernel of a single patch — its copy of the patch sat in 168 bytes of scratch per lane until its event lambda read
    # the kernel-argument patch instead (12.9 MB of HBM traffic per launch at 65 536 voices, now 5.6: profiles/r04/c2_summary.json)
    pool = {k: v for k, v in md.items() if "s2r_pool_kernelILi1ELb0E" in k}
    assert len(pool) == 1, sorted(md)
    for name, v in pool.items():
        assert v[".private_segment_fixed_size"] == 0 and v[".vgpr_spill_count"] == 0 and v[".vgpr_count"] <= 512, (name, v)
    general = {k: v for k, v in md.items() if "s2r_render_general_kernelILi" in k and "ELb0ELi256EEE" in k}
    assert len(general) == 4, sorted(md)
    for name, v in general.items():
        assert v[".private_segment_fixed_size"] == 0 and v[".vgpr_spill_count"] == 0, (name, v)
