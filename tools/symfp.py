"""tools/symfp.py -- symbolic execution of the floating-point dataflow of one gfx950 kernel's assembly (hipcc -S output).

Why: whether two rasterizers decide a (pixel, Gaussian) pair the same way at alpha = 1/255 depends on the BITS of the Gaussian's
projected centre and conic (profiles/r06_parity_report.txt: 94 % of the threshold flips against the reference build come from
per-Gaussian state that differs in its last bits, not from how alpha is evaluated).  IEEE operations are deterministic, so two
kernels produce the same bits iff they apply the same tree of operations (which product is fused into which sum, which
division is a true division) to the same inputs.  This tool prints that tree for every value a kernel stores, from its
assembly, so that the tree a compiler built for one source can be restated operation by operation in another (explicit fmaf,
contraction off) and the restatement CHECKED against it without a GPU:

    python tools/symfp.py kernel.s 'mangled_name_prefix' [--kargs 0x58=view,0x60=proj,...] [--stores]

Handles what the preprocess kernels contain: scalar / packed f32 arithmetic with op_sel / neg modifiers, the IEEE division
sequence (v_div_scale / v_rcp / v_div_fmas / v_div_fixup -> div), f64 conversions and arithmetic, moves; anything else
becomes an opaque node.  Control flow is ignored (instructions are executed in program order): good for straight-line
per-thread math with early exits, which is what is looked at here.
"""
from __future__ import annotations

import re
import sys

COMM = {"mul", "add", "max", "min", "mul64", "add64"}
BY_LINE = set()   # --byline=L1,L2: loads at these lines are named load@LINE[i] (several loads of one array with computed addresses)


def canon(e):
    """Canonical form: operands of commutative operations sorted (bitwise the same result either way)."""
    if not isinstance(e, tuple):
        return e
    op = e[0]
    args = [canon(a) for a in e[1:]]
    if op == "mul" and ("const", "2.0") in args:      # 2 x == x + x bit for bit; compilers pick either
        x = args[1] if args[0] == ("const", "2.0") else args[0]
        return ("add", x, x)
    if op in COMM:
        args = sorted(args, key=repr)
    elif op in ("fma", "fma64"):
        ab = sorted(args[:2], key=repr)
        args = ab + [args[2]]
    return (op, *args)


def show(e, depth=0):
    if not isinstance(e, tuple):
        return str(e)
    op = e[0]
    if op == "in":
        return e[1]
    if op == "const":
        return e[1]
    if op == "neg":
        return "-" + show(e[1])
    return op + "(" + ", ".join(show(a) for a in e[1:]) + ")"


def neg(e):
    if isinstance(e, tuple) and e[0] == "neg":
        return e[1]
    return ("neg", e)


class Machine:
    def __init__(self, kargs):
        self.r = {}          # 'v12' / 's3' -> expr
        self.kargs = kargs   # offset -> name
        self.stores = []     # (line, text, [exprs])
        self.loads = []

    def get(self, name):
        return self.r.get(name, ("in", name + "?"))

    # ---- operand parsing: -> list of 32-bit component expressions (1 for scalars, n for ranges)
    def operand(self, tok):
        tok = tok.strip()
        negate = False
        if tok.startswith("-") and not re.match(r"^-[0-9.]", tok):
            negate, tok = True, tok[1:]
        absolute = False
        if tok.startswith("|") and tok.endswith("|"):
            absolute, tok = True, tok[1:-1]
        m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok)
        if m:
            vals = [self.get(f"{m.group(1)}{i}") for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        elif re.match(r"^[vs]\d+$", tok):
            vals = [self.get(tok)]
        elif tok in ("vcc", "exec", "off", "vcc_lo", "vcc_hi"):
            vals = [("in", tok)]
        else:
            vals = [("const", self.literal(tok))]
        if absolute:
            vals = [("abs", v) for v in vals]
        if negate:
            vals = [neg(v) for v in vals]
        return vals

    @staticmethod
    def literal(tok):
        if tok.startswith("0x"):
            import struct
            v = int(tok, 16)
            if v <= 0xffffffff:
                f = struct.unpack("<f", struct.pack("<I", v))[0]
                return f"{f!r}f[{tok}]"
            return tok
        return tok

    def set(self, tok, vals):
        m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok)
        if m:
            for i, v in zip(range(int(m.group(2)), int(m.group(3)) + 1), vals):
                self.r[f"{m.group(1)}{i}"] = v
        else:
            self.r[tok] = vals[0]

    def pair64(self, tok):
        """A 64-bit operand (f64 / u64) as ONE expression: kept in the low register of the pair."""
        m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok.strip())
        if m:
            return self.get(f"{m.group(1)}{m.group(2)}")
        return self.operand(tok)[0]

    def set64(self, tok, e):
        m = re.match(r"^([vs])\[(\d+):(\d+)\]$", tok.strip())
        self.r[f"{m.group(1)}{m.group(2)}"] = e
        self.r[f"{m.group(1)}{m.group(3)}"] = ("hi", e)


def parse_mods(rest):
    mods = {}
    for key in ("op_sel_hi", "op_sel", "neg_lo", "neg_hi"):
        m = re.search(key + r":\[([0-9,]+)\]", rest)
        if m:
            mods[key] = [int(x) for x in m.group(1).split(",")]
            rest = rest.replace(m.group(0), "")
    return mods, rest


def run(lines, kargs, inject=None):
    """inject: {line number: {register: leaf name}} -- after that line the registers are replaced by named inputs (cuts a long
    dataflow into stages whose interface values become leaves)."""
    M = Machine(kargs)
    kbase = {}   # 's20' (low register of a pointer pair) -> karg offset
    inject = inject or {}
    pending = None
    for ln, raw in lines:
        if pending is not None:
            for reg, name in pending.items():
                M.r[reg] = ("in", name)
            pending = None
        if ln in inject:
            pending = inject[ln]
        text = raw.split(";")[0].strip()
        if not text or text.endswith(":") or text.startswith("."):
            continue
        parts = text.split(None, 1)
        op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        mods, rest2 = parse_mods(rest)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest2.strip()) if o.strip()]
        # strip trailing non-operand tokens like 'offset:16'
        offset = 0
        clean = []
        for o in ops:
            mo = re.search(r"offset:(-?\d+)", o)
            if mo:
                offset = int(mo.group(1))
                o = o[:mo.start()].strip()
            if o:
                clean.append(o)
        ops = clean
        base = re.sub(r"_e32$|_e64$", "", op)
        # pointer provenance survives only the instructions that move or offset a pointer; every other write clears it
        if ops and not base.startswith(("s_load", "s_mov_b", "v_mov_b", "v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64",
                                        "s_add_u32", "s_addc_u32", "global_store", "global_load", "s_waitcnt", "s_cmp", "v_cmp", "s_cbranch")):
            md = re.match(r"^([vs])\[(\d+):(\d+)\]$", ops[0])
            if md:
                for i in range(int(md.group(2)), int(md.group(3)) + 1):
                    kbase.pop(f"{md.group(1)}{i}", None)
            else:
                kbase.pop(ops[0], None)
        try:
            # ---------------- scalar loads from the kernel arguments / through argument pointers
            if base.startswith("s_load_dword"):
                n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[base]
                dst, src, off = ops[0], ops[1], int(ops[2], 16) if ops[2].startswith("0x") else int(ops[2])
                lo = re.match(r"^s\[(\d+):", src)
                src_lo = "s" + lo.group(1)
                d0 = int(re.match(r"^s\[?(\d+)", dst).group(1))
                if src_lo in ("s0",) or (src_lo in kbase and kbase[src_lo] == "KARG"):
                    for i in range(n):
                        M.r[f"s{d0 + i}"] = ("in", f"karg[{off + 4 * i:#x}]")
                        kbase[f"s{d0 + i}"] = off + 4 * i if (off + 4 * i) % 8 == 0 else None
                elif src_lo in kbase and kbase[src_lo] is not None:
                    name = kargs.get(kbase[src_lo], f"p{kbase[src_lo]:#x}")
                    for i in range(n):
                        M.r[f"s{d0 + i}"] = ("in", f"{name}[{(off + 4 * i) // 4}]")
                        kbase.pop(f"s{d0 + i}", None)
                else:
                    for i in range(n):
                        M.r[f"s{d0 + i}"] = ("in", f"sload@{ln}[{i}]")
                        kbase.pop(f"s{d0 + i}", None)
                continue
            if base in ("s_add_u32", "s_addc_u32") and len(ops) == 3 and ops[1] in ("s0", "s1"):
                kbase[ops[0]] = "KARG" if ops[1] == "s0" else None   # s[34:35] = kernarg + const (implicit args): not followed
                continue
            if base.startswith("global_load"):
                n = {"global_load_dword": 1, "global_load_dwordx2": 2, "global_load_dwordx3": 3, "global_load_dwordx4": 4,
                     "global_load_ushort": 1, "global_load_ubyte": 1}.get(base, 1)
                dst = ops[0]
                sbase = ops[2] if len(ops) > 2 else "off"
                lo = re.match(r"^s\[(\d+):", sbase)
                d0 = int(re.match(r"^v\[?(\d+)", dst).group(1))
                va = re.match(r"^v\[(\d+):", ops[1]) if len(ops) > 1 else None
                if ln in BY_LINE:
                    va = lo = None
                if lo and kbase.get("s" + lo.group(1)) not in (None, "KARG"):
                    name = kargs.get(kbase["s" + lo.group(1)], f"p{kbase['s' + lo.group(1)]:#x}")
                    rowed = ops[1].startswith("v") and not re.match(r"^v\d+$", ops[1]) is None and M.r.get(ops[1], ("const", "0")) != ("const", "0")
                    vals = [("in", f"{name}[{'+' if rowed else ''}{(offset + 4 * i) // 4}]") for i in range(n)]
                elif va and kbase.get("v" + va.group(1)) not in (None, "KARG"):
                    name = kargs.get(kbase["v" + va.group(1)], f"p{kbase['v' + va.group(1)]:#x}")
                    vals = [("in", f"{name}[+{(offset + 4 * i) // 4}]") for i in range(n)]
                else:
                    vals = [("in", f"load@{ln}[{i}]") for i in range(n)]
                    M.loads.append((ln, text))
                for i, v in enumerate(vals):
                    M.r[f"v{d0 + i}"] = v
                    kbase.pop(f"v{d0 + i}", None)
                continue
            if base.startswith("global_store") or base.startswith("global_atomic"):
                M.stores.append((ln, text, M.operand(ops[1])))
                continue
            # ---------------- moves
            if base in ("v_mov_b32", "s_mov_b32", "v_mov_b64", "s_mov_b64", "v_readfirstlane_b32"):
                M.set(ops[0], M.operand(ops[1]))
                if ops[1] in kbase:
                    kbase[ops[0]] = kbase[ops[1]]
                else:
                    kbase.pop(ops[0], None)
                m2 = re.match(r"^s\[(\d+):(\d+)\]$", ops[1])
                m1 = re.match(r"^s\[(\d+):(\d+)\]$", ops[0])
                if m1 and m2 and ("s" + m2.group(1)) in kbase:
                    kbase["s" + m1.group(1)] = kbase["s" + m2.group(1)]
                continue
            if base == "v_pk_mov_b32":
                a, b = M.operand(ops[1]), M.operand(ops[2])
                sel = mods.get("op_sel", [0, 0])
                M.set(ops[0], [a[sel[0]] if len(a) > 1 else a[0], b[sel[1]] if len(b) > 1 else b[0]])
                continue
            # ---------------- scalar f32 arithmetic
            un = lambda k: M.operand(ops[k])[0]
            if base in ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_max_f32", "v_min_f32"):
                a, b = un(1), un(2)
                kind = base[2:5]
                if base == "v_sub_f32":
                    e = ("add", a, neg(b))
                elif base == "v_subrev_f32":
                    e = ("add", b, neg(a))
                else:
                    e = (kind, a, b)
                M.set(ops[0], [e])
                continue
            if base == "v_fma_f32":
                M.set(ops[0], [("fma", un(1), un(2), un(3))])
                continue
            if base == "v_fmac_f32":
                M.set(ops[0], [("fma", un(1), un(2), un(0))])
                continue
            if base == "v_fmamk_f32":   # D = S0 * K + S1
                M.set(ops[0], [("fma", un(1), un(2), un(3))])
                continue
            if base == "v_fmaak_f32":   # D = S0 * S1 + K
                M.set(ops[0], [("fma", un(1), un(2), un(3))])
                continue
            if base == "v_xor_b32" and "0x80000000" in (ops[1], ops[2]):
                other = ops[2] if ops[1] == "0x80000000" else ops[1]
                M.set(ops[0], [neg(M.operand(other)[0])])
                continue
            # ---------------- packed f32
            if base in ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"):
                nsrc = 3 if base == "v_pk_fma_f32" else 2
                srcs = [M.operand(ops[1 + i]) for i in range(nsrc)]
                sel = mods.get("op_sel", [0] * nsrc)
                selh = mods.get("op_sel_hi", [1] * nsrc)
                nlo = mods.get("neg_lo", [0] * nsrc)
                nhi = mods.get("neg_hi", [0] * nsrc)

                def pick(vals, idx):
                    return vals[idx] if len(vals) > 1 else vals[0]
                out = []
                for which, (se, ne) in enumerate(((sel, nlo), (selh, nhi))):
                    xs = []
                    for i in range(nsrc):
                        v = pick(srcs[i], se[i])
                        if ne[i]:
                            v = neg(v)
                        xs.append(v)
                    kind = {"v_pk_mul_f32": "mul", "v_pk_add_f32": "add", "v_pk_fma_f32": "fma"}[base]
                    out.append((kind, *xs))
                M.set(ops[0], out)
                continue
            # ---------------- IEEE division: the fix-up names numerator and denominator
            if base == "v_div_fixup_f32":
                M.set(ops[0], [("div", un(3), un(2))])
                continue
            if base in ("v_div_scale_f32", "v_div_fmas_f32", "v_rcp_f32"):
                d = ops[0]
                M.set(d, [("divstep", base)])
                continue
            # ---------------- f64
            if base == "v_cvt_f64_f32":
                M.set64(ops[0], ("f64", un(1)))
                continue
            if base in ("v_cvt_f64_i32", "v_cvt_f64_u32"):
                M.set64(ops[0], ("f64i", un(1)))
                continue
            if base == "v_cvt_f32_f64":
                M.set(ops[0], [("f32", M.pair64(ops[1]))])
                continue
            if base in ("v_add_f64", "v_mul_f64", "v_min_f64", "v_max_f64"):
                M.set64(ops[0], (base[2:5] + "64", M.pair64(ops[1]), M.pair64(ops[2])))
                continue
            if base == "v_fma_f64":
                M.set64(ops[0], ("fma64", M.pair64(ops[1]), M.pair64(ops[2]), M.pair64(ops[3])))
                continue
            if base in ("v_cvt_f32_i32", "v_cvt_i32_f32", "v_ceil_f32", "v_floor_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rndne_f32",
                        "v_cvt_u32_f64", "v_cvt_f32_u32", "v_trunc_f32", "v_rsq_f32"):
                M.set(ops[0], [(base[2:], M.pair64(ops[1]) if "f64" in base.split("_")[-1] else un(1))])
                continue
            if base in ("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64"):
                prov = None
                for o in (ops[2:] if base.startswith("v_mad_") else ops[1:]):   # (v_mad_*64: ops[1] is the carry-out pair)
                    mm = re.match(r"^([vs])\[(\d+):", o)
                    key = f"{mm.group(1)}{mm.group(2)}" if mm else o
                    if kbase.get(key) not in (None, "KARG"):
                        prov = kbase[key]
                mm = re.match(r"^([vs])\[(\d+):(\d+)\]$", ops[0])
                M.set(ops[0], [("opq", base)] * 2)
                if mm:
                    if prov is not None:
                        kbase[f"{mm.group(1)}{mm.group(2)}"] = prov
                    else:
                        kbase.pop(f"{mm.group(1)}{mm.group(2)}", None)
                continue
            # anything else that writes a register: opaque
            if ops and re.match(r"^[vs](\d+|\[\d+:\d+\])$", ops[0]) and not base.startswith(("s_cmp", "v_cmp", "s_cbranch", "s_waitcnt", "s_nop", "s_and_saveexec")):
                srcs = []
                for o in ops[1:]:
                    try:
                        srcs.extend(M.operand(o))
                    except Exception:
                        pass
                m = re.match(r"^([vs])\[(\d+):(\d+)\]$", ops[0])
                n = int(m.group(3)) - int(m.group(2)) + 1 if m else 1
                M.set(ops[0], [("opq", base, *srcs)] * n if n == 1 else [("opq", base + f".{i}", *srcs) for i in range(n)])
                for k in list(kbase):
                    if k == ops[0] or (m and k == f"{m.group(1)}{m.group(2)}"):
                        kbase.pop(k)
        except Exception as ex:   # keep going: an instruction this tool does not model must not hide the rest
            print(f"# line {ln}: {text}  -> {ex!r}", file=sys.stderr)
    return M


def ssa(exprs, rename=None):
    """Straight-line program for a list of (label, expr): every distinct non-leaf node once, in dependency order."""
    rename = rename or {}
    names, out = {}, []

    def relabel(e):      # leaves get their final names BEFORE the canonical operand order is fixed
        if not isinstance(e, tuple):
            return e
        if e[0] in ("in", "const"):
            t = show(e)
            return (e[0], rename.get(t, t))
        return (e[0], *[relabel(a) for a in e[1:]])

    def leaf(e):
        return show(e)

    def walk(e):
        if not isinstance(e, tuple) or e[0] in ("in", "const"):
            return leaf(e)
        if e[0] == "neg":
            return "-" + walk(e[1])
        key = repr(e)
        if key in names:
            return names[key]
        args = [walk(a) for a in e[1:]]
        nm = f"t{len(names)}"
        names[key] = nm
        out.append(f"{nm} = {e[0]}({', '.join(args)})")
        return nm
    res = [(label, walk(canon(relabel(e)))) for label, e in exprs]
    return out, res


def extract(path, prefix):
    out, on = [], False
    for i, l in enumerate(open(path), 1):
        if not on and l.startswith(prefix) and l.rstrip().split(";")[0].strip().endswith(":"):
            on = True
            continue
        if on:
            out.append((i, l.rstrip("\n")))
            if "s_endpgm" in l:
                break
    return out


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    kargs = {}
    for a in sys.argv[3:]:
        if a.startswith("--kargs"):
            for kv in a.split("=", 1)[1].split(","):
                k, v = kv.split(":")
                kargs[int(k, 16)] = v
    lines = extract(path, prefix)
    for a in sys.argv[3:]:
        if a.startswith("--byline="):
            BY_LINE.update(int(x) for x in a.split("=", 1)[1].split(","))
    inject = {}
    for a in sys.argv[3:]:
        if a.startswith("--inject="):     # --inject=LINE:v0=c0;v1=c1
            ln_, regs = a.split("=", 1)[1].split(":", 1)
            inject[int(ln_)] = dict(kv.split("=") for kv in regs.split(";"))
    M = run(lines, kargs, inject)
    want = [a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("--ssa=")]
    rename = {}
    for a in sys.argv[3:]:
        if a.startswith("--rename="):
            for kv in a.split("=", 1)[1].split(","):
                k, v = kv.split(":")
                rename[k] = v
    if want:
        sel = []
        for w in want:          # LINE:INDEX[:label]
            bits = w.split(":")
            for ln, text, vals in M.stores:
                if ln == int(bits[0]):
                    sel.append((bits[2] if len(bits) > 2 else w, vals[int(bits[1])]))
        prog, res = ssa(sel, rename)
        print("\n".join(prog))
        for label, nm in res:
            print(f"OUT {label} = {nm}")
        return
    for ln, text, vals in M.stores:
        print(f"--- line {ln}: {text}")
        for i, v in enumerate(vals):
            print(f"  [{i}] {show(canon(v))}")
    if "--loads" in sys.argv:
        for ln, text in M.loads:
            print(f"load line {ln}: {text}")


if __name__ == "__main__":
    main()
