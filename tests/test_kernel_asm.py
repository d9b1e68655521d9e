"""CPU only: properties of the generated gfx950 code that the kernels rely on but the compiler does
not promise.  The MFMAs are issued from inline asm, so the compiler pads no hazards for them and is
free to park their operands in the other register file; these checks read the device assembly.

  * no kernel spills to scratch;
  * in the hot path of the tuned kernels' main loops (header to back edge; the rare rescale blocks are
    laid out behind the back edge) no value moves between the VGPR and AGPR files (a v_accvgpr_write
    placed right in front of an unpadded MFMA would be read too early, a v_accvgpr_read right behind
    one reads a result that is not there yet);
  * no MFMA result is touched by anything but a chained MFMA before its result latency is over;
  * the main loops hold exactly the expected number of MFMAs per unrolled tile triple;
  * the destinations of the converting loads of the fp32 routes (issued from inline asm, unknown to the
    compiler as loads) are left alone until a wait for vector memory;
  * dynamic LDS stays within the 160 KiB of a CU.
"""
import importlib.util
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def asm_text():
    spec = importlib.util.spec_from_file_location("fa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.build_device_asm()
    with open(path) as f:
        return f.read()


def kernels(asm_text):
    """name -> list of instruction lines (labels kept, comments and directives dropped)."""
    out = {}
    # a nested loop's header annotation sits on a comment line of its own: pull it up onto the label
    asm_text = re.sub(r"^(\.LBB\d+_\d+:[^\n]*)\n\s*;\s*=>\s*This", r"\1 =>This", asm_text, flags=re.M)
    for m in re.finditer(r"^(_ZN9flash_hip\S*attn_fwd_kernel[^:\s]*):[^\n]*\n(.*?)^\s*s_endpgm", asm_text, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n")]
        out[m.group(1)] = [l for l in body if l and not l.startswith(";") and not l.startswith(".") or l.startswith(".LBB")]
    return out


def main_loop(lines, min_mfma=90):
    """Instruction lines of the smallest loop that holds at least `min_mfma` 16x16x32 MFMAs (the unrolled
    tile group; the tail loops with run-time ring slots hold a single tile)."""
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)   # branch targets inside a loop are no loops
        if not m:
            continue
        lab = m.group(1)
        for j in range(i + 1, len(lines)):
            if re.search(r"s_(cbranch_\w+|branch)\s+" + re.escape(lab) + r"\b", lines[j]):
                body = lines[i:j + 1]
                if sum("v_mfma_f32_16x16x32" in x for x in body) >= min_mfma and (best is None or len(body) < len(best)):
                    best = body
                break
    return best


def hot_path(loop):
    """The fall-through path of a loop: from its header to the (first) branch back to the header.  The
    compiler lays the out-of-line blocks of the loop (the rescale path) out behind that branch."""
    header = loop[0].split(":")[0]
    for j, l in enumerate(loop):
        if j and re.search(r"s_(cbranch_\w+|branch)\s+" + re.escape(header) + r"\b", l):
            return loop[:j + 1]
    return loop


def test_no_kernel_spills(asm_text):
    names = re.findall(r"\.name:\s+(_ZN9flash_hip\S*attn_fwd_kernel\S*)", asm_text)
    assert len(names) >= 20, names
    vals = re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm_text)
    assert vals and all(v == "0" for v in vals), vals
    # scalar registers may overflow into lanes of a vector register (the persistent d=128 kernel keeps
    # two sets of buffer descriptors), never into memory
    assert not re.search(r"\.private_segment_fixed_size:\s+[1-9]", asm_text), "a kernel uses scratch memory"
    for lds in re.findall(r"\.group_segment_fixed_size:\s+(\d+)", asm_text):
        assert int(lds) <= 160 * 1024


TUNED = {
    # mangled-name fragment (namespace + kernel + head-dim tile) ->
    #   (16x16x32 MFMAs in the unrolled main loop, barriers in it, 16-row query blocks per wave)
    "3m1619attn_fwd_kernel_m16ILi128E": (6 * 68, 3, 4),    # d=128 tile: 32 QK + 32 PV + 4 row sums per step, 3 tiles
    "3m1619attn_fwd_kernel_m16ILi64E": (6 * 36, 3, 4),     # d=64 tile: 16 QK + 16 PV + 4 row sums per step
    "3m1620attn_fwd_kernel_m16pILi128E": (8 * 68, 4, 4),   # persistent variant: 4 tiles per loop trip (the compiler
    "3m1620attn_fwd_kernel_m16pILi64E": (8 * 36, 4, 4),    # may rotate one step of them out of the loop)
    # eight-wave layout, two query blocks per wave: 16 QK + 16 PV + 2 row sums per step
    "4m16w20attn_fwd_kernel_m16pILi128E": (8 * 34, 4, 2),
    # 128 query rows per wave (d = 64 only), eight query blocks: 32 QK + 32 PV + 8 row sums per step
    "4m16x20attn_fwd_kernel_m16pILi64E": (8 * 72, 4, 8),
}


@pytest.mark.parametrize("frag", sorted(TUNED))
def test_tuned_main_loops_keep_operands_in_place(asm_text, frag):
    # three instantiations each: fp16 in/out, bfloat16 in/out, fp16 in with O written as fp32 (same loop
    # structure, different matrix-instruction mnemonics and epilogues); the four-wave kernels also exist
    # with fp32 K and V staged through registers ("Lb1": the same steps plus loads, conversions and LDS writes),
    # and so does the 128-rows-per-wave kernel
    ks = {n: l for n, l in kernels(asm_text).items() if frag in n}
    assert len(ks) == (4 if frag.startswith("3m16") or frag.startswith("4m16x") else 3), list(ks)
    for name, lines in ks.items():
        per_step = TUNED[frag][0] // (2 * TUNED[frag][1])
        loop = main_loop(lines, TUNED[frag][0] - per_step)
        assert loop is not None, f"{name}: main loop not found"
        hot = hot_path(loop)
        n_mfma = sum("v_mfma_f32_16x16x32" in l for l in hot)
        want = "v_mfma_f32_16x16x32_bf16" if "bf16_out" in name else "v_mfma_f32_16x16x32_f16"
        assert all(want in l for l in lines if "v_mfma_f32_16x16x32" in l), f"{name}: mixed operand kinds"
        assert n_mfma in (TUNED[frag][0], TUNED[frag][0] - per_step), (name, n_mfma)
        moves = [l for l in hot if l.startswith("v_accvgpr_")]
        d = 128 if "Li128E" in name else 64
        nq = TUNED[frag][2]
        base = nq * d * 3 // 8          # O^T: nq * d / 4 registers, Q: nq * d / 8
        assert not moves, f"{name}: {len(moves)} VGPR<->AGPR moves in the main loop, e.g. {moves[:3]}"
        if "Lb1" in name and "m16pI" in name:
            # persistent kernel on fp32 K/V: the tiles arrive through converting loads (typed, 16-bit destination)
            # in accumulation registers and go to LDS straight from there: no vector-ALU instruction in between
            loads = [l for l in hot if l.startswith("tbuffer_load_format_d16_xyzw")]
            landing = set().union(*(_acc_regs(l.split(None, 1)[1].split(",")[0]) for l in loads)) if loads else set()
            assert loads and landing, name
            writes = [l for l in hot if l.startswith("ds_write_b64")]
            assert len(writes) == len(loads), (name, len(loads), len(writes))
            assert all(_acc_regs(w.split(None, 1)[1].split(",")[1]) <= landing for w in writes), name
            assert not any(l.startswith("buffer_load_dwordx4") and " lds" not in l for l in hot), name
        assert not any("scratch_" in l or "buffer_store" in l for l in hot), name
        # one barrier per tile iteration
        assert sum(l.startswith("s_barrier") for l in hot) in (TUNED[frag][1], TUNED[frag][1] - 1), name
        # the row sums stay in their fixed accumulation registers, right behind O^T and Q
        sums = {f"a[{base + 4 * q}:{base + 4 * q + 3}]" for q in range(nq)}
        n_sum = 0
        for l in hot:
            dst = l.split(None, 1)[1].split(",")[0].strip() if l.startswith("v_mfma") else ""
            if dst.startswith("a["):
                lo = int(dst[2:].split(":")[0])
                assert lo + 3 < base or dst in sums, (name, l)   # O^T (below the row sums), or a row sum in its place
                n_sum += dst in sums
        assert n_sum in (nq * 2 * TUNED[frag][1], nq * (2 * TUNED[frag][1] - 1)), (name, n_sum)


def _acc_regs(operand_text):
    """Accumulation registers named in an operand list: a5 -> {5}, a[0:3] -> {0, 1, 2, 3}."""
    regs = set()
    for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(r) for r in re.findall(r"\ba(\d+)\b", operand_text))
    return regs


def test_every_unpadded_mfma_is_clear_of_accvgpr_writes(asm_text):
    """Anywhere in the tuned kernels: an MFMA issued without its own s_nop prefix must not follow, within
    two instructions, a v_accvgpr_write to one of the accumulation registers it names."""
    for name, lines in kernels(asm_text).items():
        if not any(f in name for f in TUNED):
            continue
        ins = [l for l in lines if not l.startswith(".LBB")]
        for i, l in enumerate(ins):
            if "v_mfma" not in l:
                continue
            prev = ins[max(0, i - 2):i]
            if any(p.startswith("s_nop") for p in prev[-1:]):
                continue
            used = _acc_regs(l.split(None, 1)[1])
            for p in prev:
                if p.startswith("v_accvgpr_write"):
                    written = _acc_regs(p.split(None, 1)[1].split(",")[0])
                    assert not (written & used), (name, prev, l)


def test_no_mfma_result_is_touched_before_its_latency_is_over(asm_text):
    """The other direction: the first instruction that is not a chained MFMA and names a register an
    MFMA writes must come at least passes + 3 wait states behind it (16x16x32: 11, 32x32x16: 19,
    4x4x4: 5), everywhere in every attention kernel.  The checker itself is exercised on a made-up
    sequence first."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mfma_hazards", os.path.join(ROOT, "tools", "mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    early = ["v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], a[0:3], v[0:3]", "v_add_f32 v20, v21, v22", "s_nop 3",
             "v_exp_f32_e32 v30, v1"]
    assert [(w, n) for _, _, w, n in mod.result_hazards(early)] == [(5, 11)]
    late = early[:1] + ["v_add_f32 v20, v21, v22"] * 11 + early[-1:]
    assert mod.result_hazards(late) == []
    as_operand = early[:1] + ["v_mfma_f32_4x4x4_16b_f16 a[0:3], v[0:1], v[4:5], a[0:3]"]
    assert len(mod.result_hazards(as_operand)) == 1
    # across a loop's back edge: the matrix instruction at the bottom of the loop, its reader at the header
    loop = [".LBB0_1: ; Loop Header", "v_exp_f32_e32 v30, v1", "v_add_f32 v20, v21, v22", "s_nop 2",
            "v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], a[0:3], v[0:3]", "s_add_i32 s0, s0, 1",
            "s_cbranch_scc1 .LBB0_1", "s_nop 15", "s_endpgm"]
    assert [(w, n) for _, _, w, n in mod.result_hazards(loop)] == [(2, 11)]
    padded = loop[:5] + ["s_nop 9"] + loop[5:]
    assert mod.result_hazards(padded) == []
    # into an out-of-line block behind an unconditional branch (only reachable by following the branch)
    far = ["v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], a[0:3], v[0:3]", "s_cbranch_vccnz .LBB0_9", "s_nop 15",
           "s_branch .LBB0_8", ".LBB0_9:", "v_mul_f32 v1, v1, v5", "s_branch .LBB0_8", ".LBB0_8:", "s_endpgm"]
    assert [(w, n) for _, _, w, n in mod.result_hazards(far)] == [(1, 11)]
    names = 0
    for name, lines in mod.kernels(asm_text).items():
        names += 1
        found = mod.result_hazards(lines)
        assert not found, (name, found[:4])
    assert names >= 16


def test_lds_dma_base_register_is_set_and_used_undisturbed(asm_text):
    """Every global->LDS load takes its LDS base from M0, which the kernels write themselves from inline
    asm: between that write and the load there must be at least one wait state and no other write of M0
    (the compiler does not know the register is live there)."""
    for name, lines in kernels(asm_text).items():
        ins = [l for l in lines if not l.startswith(".LBB")]
        loads = [i for i, l in enumerate(ins) if l.startswith("buffer_load") and l.rstrip().endswith("lds")]
        if "Lb1" in name:      # fp32 K/V: staged through registers, no global->LDS loads at all
            assert not loads, name
        elif "m16" in name:
            assert loads, name
        for i in loads:
            j = i - 1
            while j >= 0 and not re.match(r"s_\w+\s+m0\b", ins[j]):
                assert "m0" not in ins[j].split(None, 1)[0], (name, ins[j])
                j -= 1
            assert j >= 0 and i - j >= 2 and i - j <= 12, (name, ins[max(0, i - 6):i + 1])


def _regs(text):
    used = set()
    for c, a, b in re.findall(r"\b([av])\[(\d+):(\d+)\]", text):
        used.update((c, r) for r in range(int(a), int(b) + 1))
    for c, a in re.findall(r"\b([av])(\d+)\b", text):
        used.add((c, int(a)))
    return used


def test_converting_loads_land_undisturbed(asm_text):
    """The fp32 routes load K, V and Q with typed buffer loads issued from inline asm, which the compiler
    does not know to be loads: it is free to copy their destination registers right behind them, before the
    data has arrived (it did, wherever it ran out of registers of the landing class).  Between every such
    load and the next s_waitcnt that names vmcnt nothing may read or write its destination."""
    pat = re.compile(r"tbuffer_load_format_d16_xyzw ([av])\[(\d+):(\d+)\]")
    seen = 0
    for name, lines in kernels(asm_text).items():
        ins = [l for l in lines if not l.startswith(".LBB")]
        for k, l in enumerate(ins):
            m = pat.match(l)
            if not m:
                continue
            seen += 1
            dst = {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
            for t in range(k + 1, len(ins)):
                x = ins[t]
                if x.startswith("s_waitcnt") and "vmcnt" in x:
                    break
                assert not (_regs(x) & dst), (name, l, x)
            else:
                raise AssertionError((name, l, "no wait behind this load"))
    assert seen >= 200   # the persistent and the one-item kernels of both head-dim tiles


def test_no_mfma_reads_a_register_written_just_before_it(asm_text):
    """A vector instruction that writes a source register of an MFMA within the three instructions in
    front of it (without s_nop padding in between) is read too early by the matrix pipe."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mfma_hazards", os.path.join(ROOT, "tools", "mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, lines in mod.kernels(asm_text).items():
        if not any(f in name for f in TUNED):
            continue
        found = mod.hazards(lines, window=3)
        assert not found, (name, found[:4])


def test_no_memory_instruction_reads_a_scalar_register_a_vector_instruction_just_wrote(asm_text):
    """Buffer loads and stores read their descriptor and scalar offset from scalar registers; one that a
    vector instruction (v_readfirstlane ...) wrote needs five wait states first.  The compiler pads the
    memory instructions it issues itself, not the ones the kernels issue from inline asm."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mfma_hazards", os.path.join(ROOT, "tools", "mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = ["v_readfirstlane_b32 s8, v247", "s_lshl_b32 s8, s8, 1", "v_mul_f32 v1, v1, v5",
             "buffer_load_dwordx4 a[60:63], v197, s[36:39], s8 offen"]
    assert [w for _, _, w in mod.scalar_operand_hazards(fresh)] == [2]
    assert mod.scalar_operand_hazards(fresh[:3] + ["s_nop 2"] + fresh[3:]) == []
    desc = ["v_readfirstlane_b32 s38, v3", "s_nop 0", "buffer_load_dwordx4 v[0:3], v9, s[36:39], 0 offen lds"]
    assert len(mod.scalar_operand_hazards(desc)) == 1
    names = 0
    for name, lines in mod.kernels(asm_text).items():
        names += 1
        found = mod.scalar_operand_hazards(lines)
        assert not found, (name, found[:4])
    assert names >= 16
