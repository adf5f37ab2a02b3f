"""Code generation for tools/airgen/dsl.py machines.

emit_cpp  -> dvt_circuits_amd/csrc/gen/air_<machine>.inc : C++ templates (one
             struct per chip) used by the gfx950 quotient / permutation kernels
             (T = Fp) and by the host verifier (T = Fp4).  Constants are emitted
             in Montgomery form.
emit_c    -> oracle/gen/air_<machine>.c : plain C (canonical field, `%`
             arithmetic from oracle/field.h) for the CPU oracle.
"""
from .dsl import Expr, P

R = (1 << 32) % P
WHEN_ID = {"all": 0, "first": 1, "last": 2, "trans": 3}
SCOPE_ID = {"local": 0, "global": 1}


class Names:
    """v<n> names in order of first use: the generated text must not depend on Expr.id, a process-global counter"""

    def __init__(self):
        self.m = {}

    def __call__(self, e):
        n = self.m.get(e.id)
        if n is None:
            n = self.m[e.id] = f"v{len(self.m)}"
        return n


def topo(roots):
    order, seen = [], set()
    stack = [(r, False) for r in reversed(roots)]
    while stack:
        e, done = stack.pop()
        if done:
            order.append(e)
            continue
        if e.id in seen:
            continue
        seen.add(e.id)
        stack.append((e, True))
        if e.op not in ("const", "var"):
            for a in reversed(e.args):
                if a.id not in seen:
                    stack.append((a, False))
    return order


def n_plain(ch):
    """constraints that are not coefficient constraints of a polynomial identity (those come last)"""
    rels = getattr(ch, "poly_rels", [])
    if not rels:
        return len(ch.constraints)
    at = rels[0].first
    for r in rels:
        assert r.first == at, f"{ch.name}: polynomial identities must be asserted after every other constraint"
        at += r.K
    assert at == len(ch.constraints), f"{ch.name}: polynomial identities must be asserted after every other constraint"
    return rels[0].first


def split_parts(ch):
    """contiguous groups of the chip's (plain) constraints with roughly equal numbers of expression nodes"""
    n = max(1, getattr(ch, "quotient_parts", 1))
    total = n_plain(ch)
    if n == 1 or total < 2 * n:
        return [(0, total)]
    cost = [len(topo([e])) for e, _ in ch.constraints]
    target, parts, lo, acc = sum(cost) / n, [], 0, 0
    for i, c_ in enumerate(cost):
        acc += c_
        if acc >= target * (len(parts) + 1) and len(parts) < n - 1 and i + 1 < total:
            parts.append((lo, i + 1))
            lo = i + 1
    parts.append((lo, total))
    return parts


def split_lparts(ch):
    """contiguous groups of whole LogUp batches (two interactions each)"""
    n, total = max(1, getattr(ch, "logup_parts", 1)), len(ch.interactions)
    if n == 1 or total < 4 * n:
        return [(0, total)]
    per = -(-((total + 1) // 2) // n) * 2
    return [(lo, min(lo + per, total)) for lo in range(0, total, per)]


def cname(ch):
    return "".join(w.capitalize() for w in ch.name.split("_"))


# ------------------------------------------------------------------ C++ (product)
def cpp_node(e, nm):
    n = nm(e)
    if e.op == "const":
        return f"const T {n} = C::K(0x{e.args[0] * R % P:08x}u);"
    if e.op == "var":
        kind, idx, rot = e.args
        if kind == "pub":
            return f"const T {n} = c.pub({idx});"
        return f"const T {n} = c.{kind}({idx}, {rot});"
    a = [nm(x) for x in e.args]
    if e.op == "add":
        return f"const T {n} = {a[0]} + {a[1]};"
    if e.op == "sub":
        return f"const T {n} = {a[0]} - {a[1]};"
    if e.op == "mul":
        return f"const T {n} = {a[0]} * {a[1]};"
    if e.op == "neg":
        return f"const T {n} = -{a[0]};"
    raise ValueError(e.op)


def cpp_poly_group(out, rel, nm, ind):
    """The coefficient constraints [first, first + K) of one polynomial identity, folded at once:
    sum_k alpha^(first + k) (c_k + W_(k-1) - 256 W_k) = alpha^first (C(alpha) + (alpha - 256) W(alpha)), with
    C(alpha) = sum_terms coef s A(alpha) B(alpha) and V(alpha) = sum_i alpha^i v_i for a limb vector V."""
    seen = set()

    def need(e):
        for x in topo([e]):
            if x.id not in seen:
                seen.add(x.id)
                out.append(ind + cpp_node(x, nm))
        return nm(e)

    vec_name = {}

    def vec(v):
        key = tuple(x if isinstance(x, int) else ("e", x.id) for x in v)
        if key not in vec_name:
            items = ", ".join(f"C::K(0x{x * R % P:08x}u)" if isinstance(x, int) else need(x) for x in v)
            n = f"pv{len(vec_name)}"
            out.append(f"{ind}const T l{n}[] = {{{items}}};")
            out.append(f"{ind}const Fp4 {n} = c.poly(l{n}, {len(v)});")
            vec_name[key] = n
        return vec_name[key]

    out.append(f"{ind}// polynomial identity '{rel.name}': constraints {rel.first} .. {rel.first + rel.K - 1}")
    out.append(f"{ind}Fp4 tot = Fp4::zero();")
    for coef, sexpr, a, b in rel.terms:
        sc = need(Expr.wrap(coef) * sexpr)
        va = vec(a)
        if b is None:
            out.append(f"{ind}tot += {va} * {sc};")
        else:
            out.append(f"{ind}tot += ({va} * {vec(b)}) * {sc};")
    wl = [w - rel.sel * off for w, off in zip(rel.w, rel.w_off)]
    out.append(f"{ind}tot += c.alpha_minus(256) * {vec(wl)};")
    out.append(f"{ind}c.fold_poly({rel.first}, tot);")


def emit_cpp(machine):
    out = [f"// GENERATED by tools/airgen (machine '{machine.name}') — do not edit.", "#pragma once", ""]
    out.append("namespace dvt { namespace air_%s {" % machine.name)
    for ch in machine.chips:
        nm = cname(ch)
        max_ar = max([len(i.vals) for i in ch.interactions] + [1])
        out.append(f"struct {nm} {{")
        out.append(f"    static constexpr int MAIN_W = {ch.main_width}, PREP_W = {ch.prep_width}, N_PUB = {ch.n_pub};")
        out.append(f"    static constexpr int N_CONSTRAINTS = {len(ch.constraints)}, N_INTERACTIONS = {len(ch.interactions)}, MAX_ARITY = {max_ar};")
        out.append(f'    static constexpr const char *NAME = "{ch.name}";')
        # constraints, in N_PARTS contiguous groups: the gfx950 quotient runs one kernel per group (a single kernel
        # over all of a wide chip's constraints keeps several hundred values live: 256 VGPRs + spills, one wave per SIMD);
        # every polynomial identity (dsl.Chip.assert_poly_zero) is a part of its own, folded in closed form
        parts = split_parts(ch)
        rels = getattr(ch, "poly_rels", [])
        if n_plain(ch) == 0 and rels:
            parts = []
        out.append(f"    static constexpr int N_PARTS = {len(parts) + len(rels)};")
        out.append("    template <class C> DVT_HD static void constraints(C &c) {")
        for k in range(len(parts) + len(rels)):
            out.append(f"        constraints_part<{k}>(c);")
        out.append("    }")
        out.append("    template <int PART, class C> DVT_HD static void constraints_part(C &c) {")
        out.append("        using T = typename C::T; (void)sizeof(T); (void)c;")
        for k, (lo, hi) in enumerate(parts):
            names = Names()
            out.append(f"        if constexpr (PART == {k}) {{")
            roots = [e for e, _ in ch.constraints[lo:hi]]
            for e in topo(roots):
                out.append("            " + cpp_node(e, names))
            for i in range(lo, hi):
                e, when = ch.constraints[i]
                out.append(f"            c.constraint({i}, {WHEN_ID[when]}, {names(e)});")
            out.append("        }")
        for k, rel in enumerate(rels):
            out.append(f"        if constexpr (PART == {len(parts) + k}) {{")
            cpp_poly_group(out, rel, Names(), "            ")
            out.append("        }")
        out.append("    }")
        # interactions (K4 walks all of them in one kernel); for the quotient they come in N_LPARTS groups of whole LogUp
        # batches (pairs), each expression emitted right before the first interaction that needs it
        lparts = split_lparts(ch)
        out.append(f"    static constexpr int N_LPARTS = {len(lparts)};")
        out.append("    template <class C> DVT_HD static void interactions(C &c) {")
        for k in range(len(lparts)):
            out.append(f"        interactions_part<{k}>(c);")
        out.append("    }")
        out.append("    template <int PART, class C> DVT_HD static void interactions_part(C &c) {")
        out.append("        using T = typename C::T; (void)sizeof(T); (void)c;")
        for k, (lo, hi) in enumerate(lparts):
            names = Names()
            out.append(f"        if constexpr (PART == {k}) {{")
            seen = set()
            for j in range(lo, hi):
                it = ch.interactions[j]
                for e in topo([it.mult] + it.vals):
                    if e.id not in seen:
                        seen.add(e.id)
                        out.append("            " + cpp_node(e, names))
                vals = ", ".join(names(v) for v in it.vals)
                out.append(f"            {{ const T vals[] = {{{vals}}}; c.interaction({j}, {machine.buses[it.bus]}, {it.sign}, {SCOPE_ID[it.scope]}, {names(it.mult)}, vals, {len(it.vals)}); }}")
            out.append("        }")
        out.append("    }")
        out.append("};")
        out.append("")
    out.append(f"constexpr int N_CHIPS = {len(machine.chips)};")
    out.append("}}  // namespace")
    out.append("")
    # X-macro list for dispatch tables
    out.append(f"#define DVT_AIR_{machine.name.upper()}_CHIPS(X) \\")
    out.append(" \\\n".join(f"    X({i}, dvt::air_{machine.name}::{cname(ch)})" for i, ch in enumerate(machine.chips)))
    out.append("")
    return "\n".join(out)


# ------------------------------------------------------------------ C (oracle)
def c_node(e, nm):
    n = nm(e)
    if e.op == "const":
        return f"const bb_t {n} = {e.args[0]}u;"
    if e.op == "var":
        kind, idx, rot = e.args
        if kind == "pub":
            return f"const bb_t {n} = pub[{idx}];"
        arr = {"main": "main", "prep": "prep"}[kind] + ("_n" if rot else "_l")
        return f"const bb_t {n} = {arr}[{idx}];"
    a = [nm(x) for x in e.args]
    if e.op == "add":
        return f"const bb_t {n} = bb_add({a[0]}, {a[1]});"
    if e.op == "sub":
        return f"const bb_t {n} = bb_sub({a[0]}, {a[1]});"
    if e.op == "mul":
        return f"const bb_t {n} = bb_mul({a[0]}, {a[1]});"
    if e.op == "neg":
        return f"const bb_t {n} = bb_neg({a[0]});"
    raise ValueError(e.op)


def emit_c(machine):
    out = [f"/* GENERATED by tools/airgen (machine '{machine.name}') — do not edit.",
           " * ORACLE — TEST INFRASTRUCTURE ONLY. */",
           '#include "../field.h"', '#include "../air_oracle.h"', ""]
    for ch in machine.chips:
        nm = f"{machine.name}_{ch.name}"
        out.append(f"static void {nm}_constraints(const bb_t *main_l, const bb_t *main_n, const bb_t *prep_l, const bb_t *prep_n, const bb_t *pub, bb_t *out) {{")
        out.append("    (void)main_l; (void)main_n; (void)prep_l; (void)prep_n; (void)pub;")
        names = Names()
        for e in topo([e for e, _ in ch.constraints]):
            out.append("    " + c_node(e, names))
        for i, (e, _) in enumerate(ch.constraints):
            out.append(f"    out[{i}] = {names(e)};")
        out.append("}")
        max_ar = max([len(i.vals) for i in ch.interactions] + [1])
        out.append(f"static void {nm}_interactions(const bb_t *main_l, const bb_t *main_n, const bb_t *prep_l, const bb_t *prep_n, const bb_t *pub, bb_t *mult, bb_t *vals) {{")
        out.append("    (void)main_l; (void)main_n; (void)prep_l; (void)prep_n; (void)pub; (void)mult; (void)vals;")
        roots = []
        for it in ch.interactions:
            roots.append(it.mult)
            roots += it.vals
        names = Names()
        for e in topo(roots):
            out.append("    " + c_node(e, names))
        for j, it in enumerate(ch.interactions):
            out.append(f"    mult[{j}] = {names(it.mult)};")
            for k, v in enumerate(it.vals):
                out.append(f"    vals[{j * max_ar + k}] = {names(v)};")
        out.append("}")
        whens = ", ".join(str(WHEN_ID[w]) for _, w in ch.constraints) or "0"
        out.append(f"static const uint8_t {nm}_when[] = {{{whens}}};")
        inter = ", ".join(f"{{{machine.buses[it.bus]}, {it.sign}, {SCOPE_ID[it.scope]}, {len(it.vals)}}}" for it in ch.interactions) or "{0,0,0,0}"
        out.append(f"static const orc_interaction_info {nm}_inter[] = {{{inter}}};")
        out.append("")
    out.append(f"const orc_chip_air orc_machine_{machine.name}[] = {{")
    for ch in machine.chips:
        nm = f"{machine.name}_{ch.name}"
        max_ar = max([len(i.vals) for i in ch.interactions] + [1])
        out.append(f'    {{"{ch.name}", {ch.main_width}, {ch.prep_width}, {ch.n_pub}, {len(ch.constraints)}, {len(ch.interactions)}, {max_ar}, {nm}_when, {nm}_inter, {nm}_constraints, {nm}_interactions}},')
    out.append("};")
    out.append(f"const unsigned orc_machine_{machine.name}_nchips = {len(machine.chips)};")
    out.append("")
    return "\n".join(out)


def _ident(name):
    return name.replace("[", "_").replace("]", "")


def emit_cols_header(machine):
    """C/C++ header with the column indices of every chip (used by trace generation)."""
    up = machine.name.upper()
    out = [f"// GENERATED by tools/airgen (machine '{machine.name}') - do not edit.", "#pragma once", ""]
    for cid, ch in enumerate(machine.chips):
        cu = ch.name.upper()
        out.append(f"#define {up}_CHIP_{cu} {cid}")
        out.append(f"#define {up}_{cu}_MAIN_W {ch.main_width}")
        out.append(f"#define {up}_{cu}_PREP_W {ch.prep_width}")
        for i, n in enumerate(ch.main_names):
            out.append(f"#define {up}_{cu}_{_ident(n)} {i}")
        for i, n in enumerate(ch.prep_names):
            out.append(f"#define {up}_{cu}_P_{_ident(n)} {i}")
        out.append("")
    return "\n".join(out) + "\n"
