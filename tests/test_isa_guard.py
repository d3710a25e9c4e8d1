"""CPU: a guard over the GENERATED gfx950 code of the hot kernels (no GPU needed: `make asm` emits the device assembly
with the very flags of the shipped objects).

The kernels own literal accumulator registers (a[0:127] / a[0:255]), count LDS / MFMA wait states by hand and rely on
hipcc keeping out of their way; a clobber list says "this statement destroys a12", it does not reserve a12.  What
stands between a compiler bump and silently wrong O / dQ / dK / dV is therefore checked here, on every build:

  * no scratch, no VGPR spill in any product kernel;
  * in the kernels that name their AGPRs, no compiler-generated instruction touches the accumulator file -- every
    `a[..]` / `v_accvgpr_*` sits between ;;#ASMSTART and ;;#ASMEND;
  * the hot loops hold exactly the MFMAs the algorithm needs (a dropped or duplicated stage shows up here);
  * no `s_nop` inside the MFMA stages of the forward (each is an issue slot the loop cannot afford), and the
    register budget of the two-waves-per-SIMD kernels stays within 128 + 128.

fa2_version() carries the compiler version so a report names the toolchain that built the library."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda_flashattention_amd", "csrc")


@pytest.fixture(scope="module")
def asm():
    subprocess.check_call(["make", "-s", "-j", "4", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in ("fa2_fwd1_bf16", "fa2_bwd_bf16", "fa2_bwd_fused", "fa2_fwd_fp8", "fa2_f32", "fa2_util", "fa1_f32"):
        out[f] = open(os.path.join(CSRC, "_obj", f + ".s")).read()
    return out


def _kernels(text):
    """name -> dict(body=[lines], meta={...}) for every kernel of one .s file."""
    ks = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
        ks[m.group(1)] = {"body": m.group(2).split("\n")}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", text, flags=re.S):
        body = m.group(2)
        name = re.search(r"\.name:\s+(\S+)", body).group(1)
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", body).group(1))
        ks[name]["meta"] = dict(agpr=int(m.group(1)), total=g("vgpr_count"), scratch=g("private_segment_fixed_size"),
                                vgpr_spill=g("vgpr_spill_count"), sgpr=g("sgpr_count"))
    return ks


def _split_asm(body):
    """(compiler_lines, asm_blocks): instructions outside / inside ;;#ASMSTART .. ;;#ASMEND."""
    outside, blocks, cur = [], [], None
    for l in body:
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            cur = []
        elif s.startswith(";;#ASMEND"):
            blocks.append(cur)
            cur = None
        elif cur is not None:
            cur.append(s)
        elif s and not s.startswith(";") and not s.startswith("."):
            outside.append(s)
    return outside, blocks


def _main_loop(body):
    """Lines of the hot loop: among the loops (label .. last backward branch to it) holding at least half as many MFMAs
    as the richest one, the shortest."""
    cands = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        tails = [j for j in range(i, len(body)) if re.match(r"\s+s_c?branch\S*\s+" + re.escape(m.group(1)) + r"\b", body[j])]
        if tails:
            cands.append((sum("v_mfma" in t for t in body[i:tails[-1]]), tails[-1] - i, i, tails[-1]))
    top = max(c[0] for c in cands)
    _, _, lo, hi = min((c for c in cands if c[0] * 2 >= top), key=lambda c: c[1])
    return body[lo:hi]


PRODUCT = ("fa2_fwd1_bf16", "fa2_bwd_bf16", "fa2_bwd_fused", "fa2_fwd_fp8", "fa2_f32", "fa2_util", "fa1_f32")


def test_no_scratch_no_spill(asm):
    bad = []
    for f in PRODUCT:
        for name, k in _kernels(asm[f]).items():
            # scratch memory: never.  VGPR "spills" into free AGPRs are tolerated only in the didactic FA1 kernel (no
            # hand-named registers there; its two 128-float rows per lane simply do not fit 256 VGPRs).
            if k["meta"]["scratch"] or (k["meta"]["vgpr_spill"] and f != "fa1_f32"):
                bad.append((name, k["meta"]))
    assert not bad, bad


@pytest.mark.parametrize("file,pattern", [("fa2_fwd1_bf16", "fa2_fwd1_bf16_kernel"), ("fa2_fwd1_bf16", "fa2_fwd1x2_bf16_kernel"),
                                          ("fa2_bwd_bf16", "fa2_bwd_dq_kernel"),
                                          ("fa2_bwd_bf16", "fa2_bwd_dkdv_kernel"), ("fa2_bwd_fused", "fa2_bwd_fused_kernelILb")])
def test_accumulator_file_is_touched_by_asm_only(asm, file, pattern):
    """The kernels that name literal AGPRs: nothing hipcc generates may read, write, copy or spill an accumulator."""
    ks = {n: k for n, k in _kernels(asm[file]).items() if pattern in n}
    assert ks
    for name, k in ks.items():
        outside, blocks = _split_asm(k["body"])
        hits = [s for s in outside if re.search(r"v_accvgpr|\ba\[\d+|\ba\d+\b", s)]
        assert not hits, (name, hits[:5])
        assert any("v_mfma" in s for b in blocks for s in b)


@pytest.mark.parametrize("pattern,limit,top128,top64", [("fa2_bwd_dkdv_kernel", 60, 255, 219), ("fa2_bwd_dq_kernel", 64, 255, 251)])
def test_named_vgprs_are_the_bodies_own(asm, pattern, limit, top128, top64):
    """The backward kernels are compiled with amdgpu_num_vgpr(60 / 64): hipcc allocates the low registers only, the rest
    belong to the generated bodies (tools/gen_dkdv_body.py, tools/gen_dq_body.py).  Nothing outside the asm regions may
    name them, and the descriptor must cover the whole file (one wave per SIMD)."""
    ks = {n: k for n, k in _kernels(asm["fa2_bwd_bf16"]).items() if pattern in n}
    assert len(ks) == 4
    pat = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    for name, k in ks.items():
        outside, blocks = _split_asm(k["body"])
        for s in outside:
            for m in pat.finditer(s):
                hi = int(m.group(1)) if m.group(1) else int(m.group(3))
                assert hi < limit, (name, s)
        assert k["meta"]["total"] == 512 and k["meta"]["agpr"] == 256, (name, k["meta"])
        used = max(int(m.group(1) or m.group(3)) for b in blocks for s in b for m in pat.finditer(s))
        assert used == (top128 if "ILi128E" in name else top64), (name, used)      # VEND - 1 of the generator


def test_generated_forward_kernel(asm):
    """What fa2_forward runs for bf16 (csrc/fa2_fwd1_bf16.hip, bodies from tools/gen_fwd_body.py), in its two shapes:
    fa2_fwd1_bf16_kernel -- one wave per SIMD, 64 rows per wave, amdgpu_num_vgpr(64), 512 registers -- and
    fa2_fwd1x2_bf16_kernel -- two waves per SIMD, 32 rows per wave, amdgpu_num_vgpr(40), 128 + 128 registers (d = 64).
    Per ring buffer and key block of a tile there is a plain, a masked and a no-maxima body with QBS (KS + 2 DT) MFMAs; the bodies of a
    tile's last key block open with vmcnt(0) + s_barrier and carry the LDS-DMA pieces of the tile two ahead (32 per
    workgroup and tile); nothing outside the asm regions names a body register."""
    ks = {n: k for n, k in _kernels(asm["fa2_fwd1_bf16"]).items() if "fa2_fwd1" in n and "bf16_kernel" in n}
    assert len(ks) == 8           # d = 128 in the one-wave shape, d = 64 in the two-wave shape, each x causal x state
    pat = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    for name, k in ks.items():
        D = 128 if "ILi128E" in name else 64
        qbs = 1 if "fwd1x2" in name else 2
        nh = 2 if D == 128 else 4
        per_body = qbs * (D // 16 + 2 * (D // 32))
        own = 64 if qbs == 2 else 40
        outside, blocks = _split_asm(k["body"])
        for s in outside:
            for m in pat.finditer(s):
                hi = int(m.group(1)) if m.group(1) else int(m.group(3))
                assert hi < own, (name, s)
        assert k["meta"]["scratch"] == 0 and k["meta"]["vgpr_spill"] == 0, (name, k["meta"])
        assert k["meta"]["total"] == (512 if qbs == 2 else 256), (name, k["meta"])
        bodies = [b for b in blocks if sum("v_mfma_f32_32x32x16_bf16" in s for s in b) == per_body and len(b) > 4 * per_body]
        # the loop over whole unmasked rounds holds three rounds of 4 tiles: with maxima (a `safe` pass), first tile with and
        # the other three without (the first round), all without; the general loop 4 nh plain + 4 nh masked bodies
        assert len(bodies) == 20 * nh, (name, len(bodies))
        with_barrier = [b for b in bodies if any(s.startswith("s_barrier") for s in b)]
        assert len(with_barrier) == 20            # one per tile
        for b in with_barrier:
            assert b[0].startswith("s_waitcnt vmcnt(0)") and b[1].startswith("s_barrier")
            assert sum(s.startswith("buffer_load_dwordx4") and s.endswith(" lds") for s in b) == 32 // (8 // qbs)
        for b in bodies:
            if b not in with_barrier:
                assert not any("buffer_load" in s for s in b)
        assert sum(any(s.startswith("v_cndmask_b32") for s in b) for b in bodies) == 4 * nh      # the masked variants
        assert sum(not any(s.startswith("v_max3_f32") for s in b) for b in bodies) == 7 * nh      # the variants without maxima


def test_fused_backward_kernel(asm):
    """The single-kernel backward is compiled with amdgpu_num_vgpr(39): v39 (progress-word prefetch) and v40..v255 belong to
    the generated body (tools/gen_fused_body.py).  A unit's steps run in three loops of six bodies of 80 MFMAs (ring of three
    Q/dO buffers x two dS tiles): the first six steps, the steady-state sixes (every per-step selection decided: ~8 scalar
    instructions between two bodies instead of ~25) and the last ones; the causal form adds a loop of six masked bodies.  The
    chained form's bodies carry their four dQ stores in front of the barrier and the four running-sum loads behind it, and
    nothing touches scratch."""
    ks = {n: k for n, k in _kernels(asm["fa2_bwd_fused"]).items() if "fa2_bwd_fused_kernelILb" in n}
    assert len(ks) == 10         # <atomics>, <chain>, <chain, causal>, the two ragged instantiations of the chained forms, the
    #                              rectangular-block one (round 4: the causal ring's unmasked half blocks) and head_dim 64's four
    pat = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    for name, k in ks.items():
        chain, causal, ragged = "ILb1ELb" in name, "ILb1ELb1ELb" in name, name.split("fa2_bwd_fused_kernelILb")[1][8:9] == "1"
        outside, blocks = _split_asm(k["body"])
        for s in outside:
            for m in pat.finditer(s):
                hi = int(m.group(1)) if m.group(1) else int(m.group(3))
                assert hi < 39, (name, s)
        d64 = name.endswith("ELi64EEEvNS_9FusedArgsE")          # the fifth template argument: head_dim
        per_body, nzero = (40, 8) if d64 else (80, 16)
        assert k["meta"]["total"] == 512 and k["meta"]["agpr"] == 256, (name, k["meta"])      # (head_dim 64 names a[0:128) and v40..v227 only; the clobber lists are head_dim 128's)
        nb = 24 if (causal or ragged) else 18      # six-body loops inside the unit loop: first, steady state, last, and masked -- the
        #                                causal diagonal, or the last key block of a ragged sequence; + the accumulator-zeroing MFMAs
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in k["body"]) == nb * per_body + nzero, name
        assert not any("scratch_" in l for l in k["body"]), name
        bodies = [b for b in _split_asm(k["body"])[1] if sum("v_mfma" in s for s in b) == per_body]
        assert len(bodies) == nb
        assert sum(any(s.startswith("v_cmp_le_i32 vcc") for s in b) for b in bodies) == (6 if (causal or ragged) else 0)
        for b in bodies:
            st = [i for i, s in enumerate(b) if s.startswith("buffer_store_dwordx4")]
            ld = [i for i, s in enumerate(b) if s.startswith("buffer_load_dwordx4 v[")]          # running sums (not the LDS-DMA)
            dma = [i for i, s in enumerate(b) if s.startswith("buffer_load_dword") and s.endswith(" lds")]
            bar = [i for i, s in enumerate(b) if s.startswith("s_barrier")]
            assert len(bar) == 1 and b[bar[0] - 1].startswith("s_waitcnt vmcnt(0)")
            if chain:
                assert len(st) == 4 and len(ld) == 4 and max(st) < bar[0] < min(ld), name
                assert all(" sc1" in b[i] for i in ld)
                # the step's LDS-DMA, progress prefetch and hand-shake are inside the body: 4 + 1 DMA issues in front of the
                # barrier, the publishing store and the bounded poll loop behind it, in front of the running-sum loads
                assert len(dma) == (3 if d64 else 5) and max(dma) < bar[0]
                pub = [i for i, s in enumerate(b) if s.startswith("buffer_store_dword v39")]
                assert len(pub) == 1 and bar[0] < pub[0] < min(ld)
                assert sum(s.startswith("s_cbranch_scc1 1b") for s in b) == 1
            else:
                assert not st and not ld and not dma


def test_backward_loop_shape(asm):
    ks = _kernels(asm["fa2_bwd_bf16"])
    for D in (128, 64):
        ksteps, dts = D // 16, D // 32
        for causal in (0, 1):
            dq = next(n for n in ks if f"fa2_bwd_dq_kernelILi{D}ELb{causal}E" in n)
            # per 64-key tile and wave: S^T and dP^T 2 x 2 x ksteps each, dQ^T 2 x 2 x 2 x dts; two tiles, and the
            # masked variant of each tile is a second copy of the body
            per_tile = 8 * ksteps + 8 * dts
            n = sum("v_mfma_f32_32x32x16_bf16" in l for l in _main_loop(ks[dq]["body"]))
            assert n == 6 * per_tile, (dq, n)          # ring of three tiles, plain and masked bodies
            dk = next(n for n in ks if f"fa2_bwd_dkdv_kernelILi{D}ELb{causal}E" in n)
            # per 32-row sub-tile and wave: S', dP' 2 x ksteps each, dV^T, dK^T 2 x 2 x dts each (generated bodies)
            per_sub = 4 * ksteps + 8 * dts
            n = sum("v_mfma_f32_32x32x16_bf16" in l for l in _main_loop(ks[dk]["body"]))
            assert n == 8 * per_sub, (dk, n)      # two tiles x two sub-tiles, plain and masked bodies


def test_fp8_loop_budget(asm):
    ks = _kernels(asm["fa2_fwd_fp8"])
    for name, k in ks.items():
        if "fa2_fwd_fp8_kernel" in name:
            assert k["meta"]["agpr"] <= 128 and k["meta"]["total"] <= 256, (name, k["meta"])
            assert any("v_mfma_f32_32x32x64_f8f6f4" in l for l in _main_loop(k["body"]))


def test_generated_fp8_forward_kernel(asm):
    """The fp8 forward (csrc/fa2_fwd_fp8.hip, bodies from tools/gen_fwd_fp8_body.py): two waves per SIMD, amdgpu_num_vgpr(32),
    128 + 128 registers.  Per ring buffer and half of a tile there is a plain, a masked and a no-maxima body with 8 MFMAs
    (v_mfma_f32_32x32x64_f8f6f4); the bodies of a tile's second half open with vmcnt(0) + s_barrier and carry the wave's four
    LDS-DMA pieces of the tile two ahead; nothing outside the asm regions names a body register or an accumulator."""
    ks = {n: k for n, k in _kernels(asm["fa2_fwd_fp8"]).items() if "fa2_fwd_fp8_kernel" in n}
    assert len(ks) == 2           # causal or not
    pat = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    for name, k in ks.items():
        outside, blocks = _split_asm(k["body"])
        for s in outside:
            for m in pat.finditer(s):
                hi = int(m.group(1)) if m.group(1) else int(m.group(3))
                assert hi < 32, (name, s)
        hits = [s for s in outside if re.search(r"v_accvgpr|\ba\[\d+|\ba\d+\b", s)]
        assert not hits, (name, hits[:5])
        assert k["meta"]["scratch"] == 0 and k["meta"]["vgpr_spill"] == 0 and k["meta"]["total"] == 256, (name, k["meta"])
        bodies = [b for b in blocks if sum("v_mfma_f32_32x32x64_f8f6f4" in s for s in b) == 8 and len(b) > 100]
        # per ring buffer and half: the round with maxima, the round without, and the general loop's plain + masked bodies
        assert len(bodies) == 4 * 2 * 4, (name, len(bodies))
        with_barrier = [b for b in bodies if any(s.startswith("s_barrier") for s in b)]
        assert len(with_barrier) == 16            # one per tile
        for b in with_barrier:
            assert b[0].startswith("s_waitcnt vmcnt(0)") and b[1].startswith("s_barrier")
            assert sum(s.startswith("buffer_load_dwordx4") and s.endswith(" lds") for s in b) == 4
        for b in bodies:
            if b not in with_barrier:
                assert not any("buffer_load" in s for s in b)
        assert sum(any(s.startswith("v_cndmask_b32") for s in b) for b in bodies) == 8       # the masked variants
        assert sum(not any(s.startswith("v_max3_f32") for s in b) for b in bodies) == 8      # the no-maxima variants


@pytest.mark.parametrize("inc", ["fa2_bwd_dkdv_body.inc", "fa2_bwd_dq_body.inc", "fa2_bwd_fused_body.inc", "fa2_fwd_body.inc",
                                 "fa2_fwd_fp8_body.inc"])
def test_generated_bodies_pass_the_static_checker(inc):
    """tools/check_body.py replays every generated main-loop body twice in a row (steady state) with an in-order model of
    the LDS queue: each MFMA source delivered by an LDS read is covered by a counted lgkmcnt, no read overwrites a
    fragment before its consumers have issued, no VALU instruction touches an MFMA result too early or a register an LDS
    read is still writing, and no MFMA reads a register a VALU instruction wrote the instruction before."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_body", os.path.join(ROOT, "tools", "check_body.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    text = open(os.path.join(CSRC, inc)).read()
    names = re.findall(r"#define (FA2_\w+_[CM]?BODY_\w+) ", text)
    assert len(names) in (12, 16, 24, 36, 120)
    for n in names:
        assert cb.check(cb.body(text, n), n) == []


def test_lds_images_are_bank_conflict_free():
    """tools/lds_bank_sim.py against the gfx950 bank rules: the swizzled [rows][D] tile image (row reads and transposed reads,
    d = 128 and 64) and -- round 4 -- the single-kernel backward's dS tile (ds_write_b64 in four groups of 16 lanes against 32
    banks, transposed reads in two groups of 32 against 64): no access pattern of the hot loops conflicts, the rounds 2-3 key of
    the dS tile reproduces its two-way store conflict, and reads find what the writes stored."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_sim", os.path.join(ROOT, "tools", "lds_bank_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.main()                                   # asserts inside
    assert m.ds_tile(m.ds_key) == (1, 1) and m.ds_tile(m.ds_key_round3) == (2, 1)
    # the key the kernel uses is the one the simulator checks
    src = open(os.path.join(CSRC, "fa2_bwd_fused.hip")).read()
    assert "#define FA2_FUSED_DSKEY(row) (((row) >> 1) & 7)" in src


def test_generated_bodies_are_up_to_date(tmp_path):
    """The committed .inc files are what the generators produce (nobody edits them by hand, nobody forgets to regenerate)."""
    for gen, inc in (("gen_dkdv_body.py", "fa2_bwd_dkdv_body.inc"), ("gen_dq_body.py", "fa2_bwd_dq_body.inc"),
                     ("gen_fused_body.py", "fa2_bwd_fused_body.inc"), ("gen_fwd_body.py", "fa2_fwd_body.inc"),
                     ("gen_fwd_fp8_body.py", "fa2_fwd_fp8_body.inc")):
        out = tmp_path / inc
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", gen), "--out", str(out)], cwd=os.path.join(ROOT, "tools"),
                              stdout=subprocess.DEVNULL)
        assert out.read_text() == open(os.path.join(CSRC, inc)).read(), inc


def test_version_names_the_compiler():
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "cuda_flashattention_amd", "lib", "libfa2_mi355x.so"))
    lib.fa2_version.restype = ctypes.c_char_p
    v = lib.fa2_version().decode()
    assert "gfx950" in v and "clang" in v and re.search(r"hip \d+\.\d+", v), v
