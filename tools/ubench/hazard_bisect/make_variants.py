"""Builds edited copies of the SLP-vectorised distance loop (victim_dist.hip) as gfx950 code objects, one instruction-level
edit each, for `../mfma_pk_hazard --co out/*.co`: which edit makes the wrong values beside a bf16 MFMA kernel go away?"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(HERE, "out")


def compile_s(flags, name):
    path = os.path.join(OUT, name + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                           *flags, "-o", path, os.path.join(HERE, "victim_dist.hip")], stderr=subprocess.DEVNULL)
    return open(path).read()


def assemble(text, name):
    s, o, co = (os.path.join(OUT, name + ext) for ext in (".s", ".o", ".co"))
    open(s, "w").write(text)
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", o, "-o", co])
    os.remove(o)


def edit(text, pattern, before=None, after=None, replace=None):
    out = []
    for line in text.split("\n"):
        if re.match(pattern, line):
            if replace is not None:
                out.append(replace(line))
                continue
            if before:
                out.append("\t" + before)
            out.append(line)
            if after:
                out.append("\t" + after)
        else:
            out.append(line)
    return "\n".join(out)


PK_RE = re.compile(r"\s+v_pk_(add|mul)_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)$")


def more_vgprs(text, n=56):
    """the edits below use v50/v51 as temporaries: raise the kernel's VGPR allocation (48 -> 56)"""
    text = re.sub(r"\.amdhsa_next_free_vgpr \d+", f".amdhsa_next_free_vgpr {n}", text)
    text = re.sub(r"\.amdhsa_accum_offset \d+", f".amdhsa_accum_offset {n}", text)
    text = re.sub(r"(\.set victim_dist\.num_vgpr, )\d+", rf"\g<1>{n}", text)
    text = re.sub(r"(\.vgpr_count:\s+)\d+", rf"\g<1>{n}", text)
    return text


def src1_halves(m):
    """(register feeding the low result, register feeding the high result) of src1 under op_sel / op_sel_hi"""
    lo, hi = int(m.group(6)), int(m.group(7))
    mods = m.group(8)
    sel_lo = hi if "op_sel:[0,1]" in mods else lo
    sel_hi = lo if "op_sel_hi:[1,0]" in mods else hi
    return sel_lo, sel_hi


def no_opsel(text):
    """op_sel / op_sel_hi on src1 replaced by two v_mov_b32 into v[50:51] and a plain packed instruction"""
    out = []
    for line in text.split("\n"):
        m = PK_RE.match(line)
        if m and "op_sel" in m.group(8):
            sel_lo, sel_hi = src1_halves(m)
            rest = re.sub(r"\s*op_sel(_hi)?:\[\d,\d\]", "", m.group(8))
            out.append(f"\tv_mov_b32_e32 v50, v{sel_lo}")
            out.append(f"\tv_mov_b32_e32 v51, v{sel_hi}")
            out.append(f"\tv_pk_{m.group(1)}_f32 v[{m.group(2)}:{m.group(3)}], v[{m.group(4)}:{m.group(5)}], v[50:51]{rest}")
        else:
            out.append(line)
    return more_vgprs("\n".join(out))


def unpack(text, which=("add", "mul"), only_opsel=False):
    """packed instructions replaced by two scalar VALU instructions, everything else (registers, order, waits) unchanged"""
    out = []
    for line in text.split("\n"):
        m = PK_RE.match(line)
        if m and m.group(1) in which and (not only_opsel or "op_sel" in m.group(8)):
            d0, d1, a0, a1 = (int(m.group(i)) for i in (2, 3, 4, 5))
            b0, b1 = src1_halves(m)
            neg = "neg_lo:[0,1]" in m.group(8)
            assert d0 not in (a1, b1) or (d0 == a1 == b1 and False), line  # low result must not clobber a high source
            op = "v_mul_f32_e32" if m.group(1) == "mul" else ("v_sub_f32_e32" if neg else "v_add_f32_e32")
            out.append(f"\t{op} v{d0}, v{a0}, v{b0}")
            out.append(f"\t{op} v{d1}, v{a1}, v{b1}")
        else:
            out.append(line)
    return "\n".join(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    slp = compile_s([], "_slp")
    noslp = compile_s(["-fno-slp-vectorize"], "_noslp")
    PK = r"\s+v_pk_(add|mul|fma)_f32"
    variants = {
        "00_orig_slp": slp,
        "01_noslp": noslp,
        "02_nop1_after_pk": edit(slp, PK, after="s_nop 1"),
        "03_nop1_before_pk": edit(slp, PK, before="s_nop 1"),
        "04_nop1_after_vmov": edit(slp, r"\s+v_mov_b32", after="s_nop 1"),
        "05_nop1_before_vmov": edit(slp, r"\s+v_mov_b32", before="s_nop 1"),
        "06_nop3_after_waitcnt": edit(slp, r"\s+s_waitcnt", after="s_nop 3"),
        "07_lgkmcnt0": edit(slp, r"\s+s_waitcnt lgkmcnt\(1\)", replace=lambda l: l.replace("lgkmcnt(1)", "lgkmcnt(0)")),
        "08_nop7_after_dsread": edit(slp, r"\s+ds_read_b128", after="s_nop 7"),
        "09_nop1_after_scalar_valu": edit(slp, r"\s+v_(sub|mul|add)_f32", after="s_nop 1"),
        "10_nop1_before_dsread": edit(slp, r"\s+ds_read_b128", before="s_nop 1"),
        "11_nop7_after_pk": edit(slp, PK, after="s_nop 7"),
        "12_no_opsel_via_vmov": no_opsel(slp),
        "13_all_pk_unpacked": unpack(slp),
        "14_only_opsel_pk_unpacked": unpack(slp, only_opsel=True),
        "15_only_pk_mul_unpacked": unpack(slp, which=("mul",)),
        "16_only_pk_add_unpacked": unpack(slp, which=("add",)),
    }
    for name, text in variants.items():
        assemble(text, name)
        print(name, len(re.findall(PK, text)), "packed-f32 instructions")


if __name__ == "__main__":
    sys.exit(main())
