"""The performance invariants of the shipped gfx950 code object, pinned on the CPU (build container; skipped where the
ROCm LLVM tools are missing).

DESIGN.md's speed rests on facts no parity test sees: 39 936 / 40 960 bytes of LDS per workgroup (exactly four per CU),
no scratch, no spills, the 83 / 79-instruction hand-scheduled symbol steps of the decoder, registers pinned by name
(v220-v255) that the compiler must leave alone.  A toolchain bump that re-pads, spills or halves the occupancy would keep
every parity test green; it must fail HERE, not in a bench.  (The reference fixes its occupancy by hand too --
32-thread blocks and 16.5 KB of shared memory per block, /root/reference/src/gpu.h:9 --; here it is a derived fact.)

What is read is the code object inside gpuar_amd/lib/libgpuar_hip.so itself (the library that ships and that the GPU tests
load): its AMDGPU metadata note (llvm-readelf --notes) and its disassembly (llvm-objdump -d).
"""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpuar_amd", "lib", "libgpuar_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = {t: os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-readelf", "llvm-objdump")}

LDS_PER_CU = 160 * 1024
# kernel (demangled name prefix) -> (LDS bytes allowed, vector registers allowed, workgroups per CU the design counts on)
BUDGET = {
    "encode_kernel": (LDS_PER_CU // 4, 128, 4),          # 4 workgroups x 4 wavefronts per CU: 3 working + 1 courier wavefront per SIMD
    "encode_small_kernel": (64 * 1024, 128, 2),          # latency mode: 7 wavefronts per workgroup, two workgroups per CU
    "decode_slots_kernel": (LDS_PER_CU // 4, 256, 4),    # one wavefront per SIMD: the whole register file of a SIMD lane is its own
    "decode_stream_kernel": (LDS_PER_CU // 4, 256, 4),
}


def _need_tools():
    missing = [p for p in TOOLS.values() if not os.path.exists(p)] + ([] if shutil.which("objcopy") else ["objcopy"])
    if missing:
        pytest.skip(f"ROCm LLVM tools not installed: {missing}")
    if not os.path.exists(LIB):
        pytest.skip("gpuar_amd/lib/libgpuar_hip.so not built")


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    """The gfx950 code object out of the shipped library: (metadata by kernel, disassembly by kernel)."""
    _need_tools()
    d = tmp_path_factory.mktemp("codeobj")
    fat, elf = str(d / "fat.bin"), str(d / "gfx950.elf")
    subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, str(d / "unused.so")])
    targets = subprocess.check_output([TOOLS["clang-offload-bundler"], "--list", "--type=o", f"--input={fat}"], text=True).split()
    gfx = [t for t in targets if t.endswith("gfx950")]
    assert len(gfx) == 1, f"the library must hold exactly one device target, gfx950: {targets}"
    assert all(t.startswith("host-") or t.endswith("gfx950") for t in targets), targets          # no second architecture, no dual path
    subprocess.check_call([TOOLS["clang-offload-bundler"], "--unbundle", "--type=o", f"--input={fat}", f"--targets={gfx[0]}", f"--output={elf}"])
    notes = subprocess.check_output([TOOLS["llvm-readelf"], "--notes", elf], text=True)
    dis = subprocess.check_output([TOOLS["llvm-objdump"], "-d", "--no-show-raw-insn", elf], text=True)
    return parse_metadata(notes), parse_disassembly(dis)


def parse_metadata(notes):
    """{kernel name: {field: int}} from the amdhsa.kernels list of the metadata note (only the scalar fields)."""
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"^\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.groups()
        if re.match(r"^  - \.", line):                       # first field of the next kernel's record
            cur = {}
            kernels[len(kernels)] = cur
        if cur is None:
            continue
        cur[key] = int(val) if re.fullmatch(r"\d+", val) else val
    by_name = {}
    for rec in kernels.values():
        if "name" in rec:
            by_name[demangled(rec["name"])] = rec
    return by_name


def demangled(sym):
    m = re.match(r"_ZN5gpuar(\d+)", sym)
    return sym[len(m.group(0)):len(m.group(0)) + int(m.group(1))] if m else sym


def parse_disassembly(dis):
    """{kernel name: [instruction text, ...]} (labels and blank lines dropped)."""
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(demangled(m.group(1)), [])
            continue
        t = line.strip()
        if cur is not None and t and not t.startswith("Disassembly"):
            cur.append(re.sub(r"\s*//.*$", "", t))
    return out


def check_resources(name, rec):
    """The per-kernel resource contract; raises AssertionError with the figure that broke it."""
    lds, vgprs, per_cu = BUDGET[name]
    assert rec["group_segment_fixed_size"] <= lds, (
        f"{name}: {rec['group_segment_fixed_size']} bytes of LDS per workgroup, more than {lds}: fewer than {per_cu} workgroups fit a CU")
    assert rec["private_segment_fixed_size"] == 0, f"{name}: {rec['private_segment_fixed_size']} bytes of scratch per lane"
    assert rec["vgpr_spill_count"] == 0 and rec["sgpr_spill_count"] == 0, f"{name}: spills ({rec['vgpr_spill_count']} vector, {rec['sgpr_spill_count']} scalar)"
    assert rec["vgpr_count"] + rec.get("agpr_count", 0) <= vgprs, f"{name}: {rec['vgpr_count']} vector registers, more than {vgprs}"
    assert rec["wavefront_size"] == 64 and rec.get("uses_dynamic_stack") in ("false", False, 0)


def test_every_kernel_keeps_its_lds_registers_and_no_scratch(code_object):
    meta, _ = code_object
    for name in BUDGET:
        assert name in meta, (name, sorted(meta))
        check_resources(name, meta[name])
    for name, rec in meta.items():                         # the auxiliary kernels (scan, gather, generators, copy) as well
        assert rec["private_segment_fixed_size"] == 0 and rec["vgpr_spill_count"] == 0 and rec["sgpr_spill_count"] == 0, name
    # the figures the design quotes: not bounds but the values themselves, so a silent re-layout shows
    assert meta["encode_kernel"]["group_segment_fixed_size"] == 39936
    assert meta["decode_slots_kernel"]["group_segment_fixed_size"] == 40960 == meta["decode_stream_kernel"]["group_segment_fixed_size"]
    assert meta["encode_kernel"]["max_flat_workgroup_size"] == 256 and meta["decode_slots_kernel"]["max_flat_workgroup_size"] == 64


def test_the_resource_check_does_fail_when_a_kernel_outgrows_its_share():
    """What an edit that pushes the LDS over 40 KiB (a bigger slot ring, a fourth sums slot) or a spill would look like."""
    good = {"group_segment_fixed_size": 39936, "private_segment_fixed_size": 0, "vgpr_spill_count": 0, "sgpr_spill_count": 0,
            "vgpr_count": 112, "agpr_count": 0, "wavefront_size": 64, "uses_dynamic_stack": "false"}
    check_resources("encode_kernel", good)
    for bad in ({"group_segment_fixed_size": 40960 + 512}, {"private_segment_fixed_size": 16}, {"vgpr_spill_count": 3}, {"vgpr_count": 132}):
        with pytest.raises(AssertionError):
            check_resources("encode_kernel", dict(good, **bad))


def test_no_matrix_no_scratch_no_buffer_instructions_anywhere(code_object):
    _, dis = code_object
    for name, text in dis.items():
        ops = collections.Counter(t.split()[0] for t in text)
        bad = [op for op in ops if op.startswith(("v_mfma", "v_smfmac", "scratch_", "buffer_"))]
        assert not bad, (name, bad)                        # integer, bit-serial work: nothing here is a contraction; nothing spills


VEC = re.compile(r"^v_")
PINNED = re.compile(r"\bv(2[2-4][0-9]|25[0-5])\b|\bv\[(2[2-4][0-9]|25[0-5]):")


def step_regions(text):
    """The decoder's symbol steps in a kernel's disassembly: a step ends with the 64-bit shift of lo : off : window, the
    only v_lshlrev_b64 on v[216:217] (GPUAR_OFF_TEXT); a region = the instructions behind one such shift up to and including
    the next."""
    ends = [i for i, t in enumerate(text) if t.startswith("v_lshlrev_b64 v[216:217]")]
    return [text[a + 1:b + 1] for a, b in zip(ends, ends[1:])]


@pytest.mark.parametrize("kernel", ["decode_slots_kernel", "decode_stream_kernel"])
def test_decoder_symbol_step_keeps_its_instruction_budget(code_object, kernel):
    """The hand-scheduled step: an EVEN step -- which refills the stream window for itself and its successor, in
    all lanes alike -- is 83 vector + 4 LDS instructions, the ODD one behind it -- which moves the reader on and reads the next
    stream dword -- 79 + 5: 81 + 4.5 per symbol, the model's total counted up by one of them (rounds 4-5: 82 + 5, two scalar
    instructions that wrote exec and an s_or per symbol for the total).  A run of eight steps is ONE asm statement: between two
    steps of a run there is NOTHING -- no s_nop, no scalar bookkeeping of the compiler's (a scalar instruction costs a lone
    wavefront a slot like any other) --, and inside a step the only scalar instructions are the two lane-mask combinations of the
    depth-1 register nodes and the two waits.  Exactly two waits, each for
    a record read with one LDS operation behind it (lgkmcnt(1)) or -- the odd step's first, which has the stream read behind it too -- two
    (lgkmcnt(2)), no vector memory instruction, no s_nop
    pad, no branch, and NOTHING that writes exec inside a step.  Two loop bodies of 32 steps per kernel (the wave-uniform one
    and the one with lanes sitting blocks out)."""
    _, dis = code_object
    regions = step_regions(dis[kernel])
    assert len(regions) == 63, len(regions)                # 2 x 32 steps
    shapes = collections.Counter()
    for r in regions:
        ops = [t.split()[0] for t in r]
        shapes[(sum(1 for o in ops if VEC.match(o)), sum(1 for o in ops if o.startswith("ds_")),
                sum(1 for t in r if t.startswith(("s_waitcnt lgkmcnt(1)", "s_waitcnt lgkmcnt(2)"))),
                sum(1 for o in ops if o.startswith(("global_", "flat_"))), sum(1 for o in ops if o == "s_nop"),
                sum(1 for o in ops if o.startswith(("s_cbranch", "s_branch"))))] += 1
    middle = {k: v for k, v in shapes.items() if k[3] == 0 and k[5] == 0}          # steps without a ring phase or a loop edge behind them
    assert sum(middle.values()) >= 54, shapes
    for (valu, ds, waits, vmem, nops, branches), count in middle.items():
        assert (valu, ds) in ((79, 5), (80, 5), (83, 4), (84, 4)) and waits == 2 and nops == 0, ((valu, ds, waits, vmem, nops, branches), count)
    two = shapes.most_common(2)
    assert {k[:3] for k, _ in two} == {(79, 5, 2), (83, 4, 2)} and min(c for _, c in two) >= 20, shapes
    # per symbol (an even step and the odd one behind it): 81 vector and 4.5 LDS instructions
    assert sum(k[0] for k, _ in two) == 162 and sum(k[1] for k, _ in two) == 9, two
    # scalar instructions of a plain step: s_and_b64 + s_andn2_b64 (the register nodes' lane masks) and the two waits, nothing else
    for r in regions:
        ops = [t.split()[0] for t in r]
        if any(o.startswith(("global_", "flat_", "s_cbranch", "s_branch")) for o in ops):
            continue
        scalar = sorted(o for o in ops if o.startswith("s_"))
        assert scalar == ["s_and_b64", "s_andn2_b64", "s_waitcnt", "s_waitcnt"], scalar
    # the (even) steps behind a ring phase carry it in their region: seven vector instructions, two LDS writes, up to three loads, one more wait; where the compiler's own
    # scalar bookkeeping for the next run meets the phase's first instruction it may need one wait state (an s_nop 0 in a
    # slot a scalar instruction would take anyway) -- one, not a pad per lane mask as in the compiler's own schedule of the step
    for (valu, ds, waits, vmem, nops, branches), count in shapes.items():
        if branches == 0:
            assert nops <= (1 if vmem else 0) and valu <= 94 and ds <= 7 and vmem <= 3, (valu, ds, waits, vmem, nops)
    # the window refill is selects on a lane mask now: inside a plain step nothing writes exec (the ring phase still does)
    for r in regions:
        if not any(t.split()[0].startswith(("global_", "flat_")) for t in r):
            assert not [t for t in r if t.startswith(("s_and_saveexec", "s_or_b64 exec", "s_mov_b64 exec"))], r


@pytest.mark.parametrize("kernel", ["decode_slots_kernel", "decode_stream_kernel"])
def test_compiler_leaves_the_decoders_pinned_registers_alone(code_object, kernel):
    """v220-v223 (the stream piece in flight) and v224-v255 (four sets of eight reciprocal multipliers) are written by loads
    that are still in flight when their asm statement ends: correctness rests on the compiler never copying,
    moving or spilling them.  Every instruction that names one of them must be one of the hand-written three: the load that
    fills it, the LDS write that files the piece, the v_mul_hi_u32 that reads a multiplier."""
    _, dis = code_object
    uses = collections.Counter()
    for t in dis[kernel]:
        if PINNED.search(t):
            op = t.split()[0]
            uses[op] += 1
            if op == "global_load_dwordx4":
                assert re.match(r"global_load_dwordx4 v\[2\d\d:2\d\d\],", t), t                 # as the destination only
            elif op == "ds_write2st64_b32":
                assert re.search(r", v22[02], v22[13]", t), t                                     # as the data only
            elif op == "v_mul_hi_u32":
                assert re.match(r"v_mul_hi_u32 v\d+, v\d+, v2\d\d$", t), t                        # as the multiplier only
            else:
                raise AssertionError(f"{kernel}: the compiler touches a pinned register: {t}")
    # two loop bodies of 32 steps (two multiplies each) and four ring phases (two LDS writes, three loads each)
    assert uses["v_mul_hi_u32"] == 2 * 64 and uses["ds_write2st64_b32"] >= 2 * 4 * 2 and uses["global_load_dwordx4"] >= 2 * 4 * 3, uses


def phase_segments(text):
    segs, cur = [], []
    for t in text:
        cur.append(t)
        if t.startswith("s_barrier"):
            segs.append(cur)
            cur = []
    return segs


def test_encoder_roles_keep_their_instruction_budgets(code_object):
    """encode_kernel's three working roles, one phase (eight symbols) between two barriers each: the top
    modeler 22 vector + 8.6 LDS instructions per symbol, the low modeler 24 + 7, the coder 31 (+ 13 of rare path that a
    scalar branch jumps over) + 1.  The SIMDs issue a vector instruction in 95 % of their slots, so every instruction the
    compiler adds (a register copy, a pad) is time."""
    _, dis = code_object
    tops, lows, coders = [], [], []
    for seg in phase_segments(dis["encode_kernel"]):
        ops = [t.split()[0] for t in seg]
        valu = sum(1 for o in ops if VEC.match(o))
        lds = sum(1 for o in ops if o.startswith("ds_"))
        branches = sum(1 for o in ops if o.startswith(("s_cbranch", "s_branch")))
        tags = ops.count("v_lshlrev_b32_sdwa")               # the row tag of a symbol: one per symbol in both modelers' whole phases
        if tags == 8 and ops.count("v_bfe_i32") == 8 and branches <= 1:
            lows.append((valu, lds))
        elif tags == 8 and branches == 0 and not any(o.startswith("global_") for o in ops):
            tops.append((valu, lds))
        elif ops.count("v_mul_hi_u32") == 16 and ops.count("global_store_dword") == 16:
            coders.append((valu, lds))
    assert len(tops) >= 6 and len(lows) == 1 and len(coders) == 1, (tops, lows, coders)
    for valu, lds in tops:
        assert valu <= 8 * 22.5 and lds <= 8 * 8.75, (valu, lds)
    assert lows[0][0] <= 8 * 24.75 and lows[0][1] <= 8 * 7.25, lows
    assert coders[0][0] <= 8 * (31 + 13) and coders[0][1] <= 12, coders
