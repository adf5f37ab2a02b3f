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


def affine(e):
    """an expression of degree <= 1 as ({(kind, idx, rot): coefficient}, constant), everything mod P"""
    if e.op == "const":
        return {}, e.args[0]
    if e.op == "var":
        return {e.args: 1}, 0
    if e.op == "neg":
        t, c = affine(e.args[0])
        return {k: (-v) % P for k, v in t.items()}, (-c) % P
    (ta, ca), (tb, cb) = affine(e.args[0]), affine(e.args[1])
    if e.op in ("add", "sub"):
        sg = 1 if e.op == "add" else -1
        t = dict(ta)
        for k, v in tb.items():
            t[k] = (t.get(k, 0) + sg * v) % P
        return {k: v for k, v in t.items() if v}, (ca + sg * cb) % P
    assert e.op == "mul" and (not ta or not tb), "not affine"
    if not ta:
        return {k: v * ca % P for k, v in tb.items() if v * ca % P}, ca * cb % P
    return {k: v * cb % P for k, v in ta.items() if v * cb % P}, ca * cb % P


def _shape(it):
    """interaction as (fixed part, [per value: sorted [(kind, rot, coef, idx)], const])"""
    vals = []
    for v in it.vals:
        t, c = affine(v)
        vals.append((sorted((k[0], k[2], cf, k[1]) for k, cf in t.items()), c))
    return (it.bus, it.sign, it.scope, len(it.vals), it.mult.id), vals


def _delta(a, b):
    """index / constant shifts that turn interaction shape a into b, or None"""
    if a[0] != b[0]:
        return None
    out = []
    for (ta, ca), (tb, cb) in zip(a[1], b[1]):
        if len(ta) != len(tb) or any(x[:3] != y[:3] for x, y in zip(ta, tb)):
            return None
        out.append((tuple(y[3] - x[3] for x, y in zip(ta, tb)), (cb - ca) % P))
    return out


def find_runs(interactions, lo, hi, min_blocks=4):
    """-> list of (j0, period, blocks) covering disjoint sub-ranges of [lo, hi): `blocks` consecutive groups of `period`
    interactions that differ only by constant shifts of their column indices and constants"""
    shapes = {j: _shape(interactions[j]) for j in range(lo, hi)}
    runs, j = [], lo
    while j < hi:
        best = None
        for p in (1, 2, 3, 4):
            if j + 2 * p > hi:
                break
            d0 = [_delta(shapes[j + t], shapes[j + p + t]) for t in range(p)]
            if any(d is None for d in d0):
                continue
            n = 2
            while j + (n + 1) * p <= hi and [_delta(shapes[j + (n - 1) * p + t], shapes[j + n * p + t]) for t in range(p)] == d0:
                n += 1
            if n >= min_blocks and (best is None or n * p > best[1] * best[2]):
                best = (j, p, n, d0)
        if best:
            runs.append(best)
            j += best[1] * best[2]
        else:
            j += 1
    return runs


def cpp_run(out, interactions, run, machine, need, ind):
    """a run of interactions as a loop over its blocks"""
    j0, p, n, deltas = run
    mults = [need(interactions[j0 + t].mult) for t in range(p)]      # (defined before the loop opens)
    out.append(f"{ind}_Pragma(\"unroll 1\") for (int i_ = 0; i_ < {n}; i_++) {{   // interactions {j0} .. {j0 + p * n - 1}")
    for t in range(p):
        it = interactions[j0 + t]
        (_, vals), dl = _shape(it), deltas[t]
        items = []
        for (terms, const), (didx, dconst) in zip(vals, dl):
            parts = []
            for (kind, rot, coef, idx), d in zip(terms, didx):
                at = f"{idx} + {d} * i_" if d else f"{idx}"
                acc = f"c.pub({at})" if kind == "pub" else f"c.{kind}({at}, {rot})"
                parts.append(acc if coef == 1 else f"C::K(0x{coef * R % P:08x}u) * {acc}")
            if dconst:
                parts.append(f"c.KI((uint32_t)(({const}ull + {dconst}ull * i_) % {P}ull))")
            elif const or not parts:
                parts.append(f"C::K(0x{const * R % P:08x}u)")
            items.append(" + ".join(parts))
        out.append(f"{ind}    {{ const T vals[] = {{{', '.join(items)}}}; c.interaction({j0 + t} + {p} * i_, {machine.buses[it.bus]}, {it.sign}, {SCOPE_ID[it.scope]}, {mults[t]}, vals, {len(it.vals)}); }}")
    out.append(f"{ind}}}")


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

            def need(e):
                for x in topo([e]):
                    if x.id not in seen:
                        seen.add(x.id)
                        out.append("            " + cpp_node(x, names))
                return names(e)

            # long regular stretches (the byte-range lookups and word accesses of the wide precompile chips) become loops
            runs = {r[0]: r for r in find_runs(ch.interactions, lo, hi)} if hi - lo >= 16 else {}
            j = lo
            while j < hi:
                if j in runs:
                    cpp_run(out, ch.interactions, runs[j], machine, need, "            ")
                    j += runs[j][1] * runs[j][2]
                    continue
                it = ch.interactions[j]
                for e in [it.mult] + it.vals:
                    need(e)
                vals = ", ".join(names(v) for v in it.vals)
                out.append(f"            {{ const T vals[] = {{{vals}}}; c.interaction({j}, {machine.buses[it.bus]}, {it.sign}, {SCOPE_ID[it.scope]}, {names(it.mult)}, vals, {len(it.vals)}); }}")
                j += 1
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
        # (big chips: one function per slice of constraints / interactions — gcc's time grows much faster than linearly with
        #  the size of a function, and a chip with big-integer identities has ~10^5 expression nodes)
        args = "const bb_t *main_l, const bb_t *main_n, const bb_t *prep_l, const bb_t *prep_n, const bb_t *pub"
        unused = "    (void)main_l; (void)main_n; (void)prep_l; (void)prep_n; (void)pub;"
        CH = 64
        n_pl = n_plain(ch)
        rels = getattr(ch, "poly_rels", [])
        nslices = max(1, -(-n_pl // CH))
        split = nslices > 1 or bool(rels)
        for sl in range(nslices):
            lo, hi = sl * CH, min(n_pl, (sl + 1) * CH)
            fname = f"{nm}_constraints_{sl}" if split else f"{nm}_constraints"
            out.append(f"static void {fname}({args}, bb_t *out) {{")
            out.append(unused + " (void)out;")
            names = Names()
            for e in topo([e for e, _ in ch.constraints[lo:hi]]):
                out.append("    " + c_node(e, names))
            for i in range(lo, hi):
                out.append(f"    out[{i}] = {names(ch.constraints[i][0])};")
            out.append("}")
        # polynomial identities: the coefficient constraints AS DEFINED (dsl.Chip.assert_poly_zero), coefficient by
        # coefficient with plain convolution loops — the product folds the same constraints in closed form instead
        for ri, rel in enumerate(rels):
            out.append(f"static void {nm}_poly_{ri}({args}, bb_t *out) {{   /* identity '{rel.name}': constraints {rel.first} .. {rel.first + rel.K - 1} */")
            out.append(unused)
            names, seen = Names(), set()

            def need(e):
                for x in topo([e]):
                    if x.id not in seen:
                        seen.add(x.id)
                        out.append("    " + c_node(x, names))
                return names(e)

            out.append(f"    bb_t c[{rel.K}];")
            out.append(f"    for (int k = 0; k < {rel.K}; k++) c[k] = 0;")
            for ti, (coef, sexpr, a, b) in enumerate(rel.terms):
                sc = need(Expr.wrap(coef) * sexpr)
                av = ", ".join(f"{x % P}u" if isinstance(x, int) else need(x) for x in a)
                bv = ", ".join(f"{x % P}u" if isinstance(x, int) else need(x) for x in b) if b is not None else ""
                out.append(f"    {{ const bb_t a[] = {{{av}}};")
                if b is None:
                    out.append(f"      for (int i = 0; i < {len(a)}; i++) c[i] = bb_add(c[i], bb_mul({sc}, a[i])); }}")
                else:
                    out.append(f"      const bb_t b[] = {{{bv}}};")
                    out.append(f"      for (int i = 0; i < {len(a)}; i++) for (int j = 0; j < {len(b)}; j++) c[i + j] = bb_add(c[i + j], bb_mul({sc}, bb_mul(a[i], b[j]))); }}")
            wv = ", ".join(need(w - rel.sel * off) for w, off in zip(rel.w, rel.w_off))
            out.append(f"    const bb_t w[] = {{{wv}, 0u}};")
            out.append(f"    for (int k = 0; k < {rel.K}; k++) out[{rel.first} + k] = bb_sub(bb_add(c[k], k ? w[k - 1] : 0u), bb_mul(256u, w[k]));")
            out.append("}")
        if split:
            out.append(f"static void {nm}_constraints({args}, bb_t *out) {{")
            for sl in range(nslices):
                out.append(f"    {nm}_constraints_{sl}(main_l, main_n, prep_l, prep_n, pub, out);")
            for ri in range(len(rels)):
                out.append(f"    {nm}_poly_{ri}(main_l, main_n, prep_l, prep_n, pub, out);")
            out.append("}")
        max_ar = max([len(i.vals) for i in ch.interactions] + [1])
        ICH = 64
        islices = max(1, -(-len(ch.interactions) // ICH))
        for sl in range(islices):
            lo, hi = sl * ICH, min(len(ch.interactions), (sl + 1) * ICH)
            fname = f"{nm}_interactions" if islices == 1 else f"{nm}_interactions_{sl}"
            out.append(f"static void {fname}({args}, bb_t *mult, bb_t *vals) {{")
            out.append(unused + " (void)mult; (void)vals;")
            roots = []
            for it in ch.interactions[lo:hi]:
                roots.append(it.mult)
                roots += it.vals
            names = Names()
            for e in topo(roots):
                out.append("    " + c_node(e, names))
            for j in range(lo, hi):
                it = ch.interactions[j]
                out.append(f"    mult[{j}] = {names(it.mult)};")
                for k, v in enumerate(it.vals):
                    out.append(f"    vals[{j * max_ar + k}] = {names(v)};")
            out.append("}")
        if islices > 1:
            out.append(f"static void {nm}_interactions({args}, bb_t *mult, bb_t *vals) {{")
            for sl in range(islices):
                out.append(f"    {nm}_interactions_{sl}(main_l, main_n, prep_l, prep_n, pub, mult, vals);")
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


def emit_rels_header(machine):
    """C++ tables describing every polynomial identity (dsl.Chip.assert_poly_zero) of the machine, for the trace
    builder: it fills the quotient q and the carries w of a row from the row's other cells (rv32_bigops.hip).  DVT_RELS_Q
    (polyrel.h) places the tables in host memory in the host pass and in device memory in the device pass."""
    up = machine.name.upper()
    out = [f"// GENERATED by tools/airgen (machine '{machine.name}') - do not edit.", "#pragma once", '#include "../polyrel.h"', ""]
    out.append(f"namespace dvt {{ namespace rels_{machine.name} {{")
    for ch in machine.chips:
        rels = getattr(ch, "poly_rels", [])
        if not rels:
            continue
        cu = ch.name
        for ri, rel in enumerate(rels):
            pre = f"{cu}_{ri}"

            def col_of(e):
                assert e.op == "var" and e.args[0] == "main" and e.args[2] == 0, "limb vectors of a polynomial identity are plain main columns"
                return e.args[1]

            def vec(tag, v):
                if v is None:
                    return "{nullptr, nullptr, 0}"
                if all(isinstance(x, int) for x in v):
                    out.append(f"DVT_RELS_Q uint8_t {pre}_{tag}c[] = {{{', '.join(str(x) for x in v)}}};")
                    return f"{{nullptr, {pre}_{tag}c, {len(v)}}}"
                out.append(f"DVT_RELS_Q int16_t {pre}_{tag}v[] = {{{', '.join(str(col_of(x)) for x in v)}}};")
                return f"{{{pre}_{tag}v, nullptr, {len(v)}}}"

            rows = []
            for ti, (coef, sexpr, a, b) in enumerate(rel.terms[:-1]):      # (the last term is - sel q modulus)
                rows.append(f"    {{{coef}, {col_of(sexpr)}, {vec(f't{ti}a', a)}, {vec(f't{ti}b', b)}}},")
            out.append(f"DVT_RELS_Q PolyTerm {pre}_terms[] = {{")
            out += rows
            out.append("};")
            nq = len(rel.q)
            out.append(f"DVT_RELS_Q int16_t {pre}_q[] = {{{', '.join(str(col_of(x)) for x in rel.q)}}};")
            if all(isinstance(b, int) for b in rel.modulus):
                pinv = pow(sum(b << (8 * i) for i, b in enumerate(rel.modulus)), -1, 1 << (8 * nq))
                out.append(f"DVT_RELS_Q uint8_t {pre}_mod[] = {{{', '.join(str(x) for x in rel.modulus)}}};")
                out.append(f"DVT_RELS_Q uint8_t {pre}_pinv[] = {{{', '.join(str((pinv >> (8 * i)) & 0xFF) for i in range(nq))}}};")
                out.append(f"DVT_RELS_Q int16_t *const {pre}_modv = nullptr;")
            else:       # the modulus is read from the row (UINT256_MUL): the solver divides
                out.append(f"DVT_RELS_Q uint8_t *const {pre}_mod = nullptr, *const {pre}_pinv = nullptr;")
                out.append(f"DVT_RELS_Q int16_t {pre}_modv[] = {{{', '.join(str(col_of(x)) for x in rel.modulus)}}};")
            out.append(f"DVT_RELS_Q int16_t {pre}_w[] = {{{', '.join(str(col_of(x)) for x in rel.w_lo)}}};")
            if rel.w_top:
                out.append(f"DVT_RELS_Q int16_t {pre}_wb[] = {{{', '.join(str(col_of(x)) for x in rel.w_top)}}};")
            out.append(f"DVT_RELS_Q int32_t {pre}_woff[] = {{{', '.join(str(x) for x in rel.w_off)}}};")
        out.append(f"DVT_RELS_Q PolyRelDesc {cu}[] = {{")
        for ri, rel in enumerate(rels):
            pre = f"{cu}_{ri}"
            wb = f"{pre}_wb" if rel.w_top else "nullptr"
            out.append(f'    {{"{rel.name}", {len(rel.terms) - 1}, {pre}_terms, {rel.K}, {pre}_q, {len(rel.q)}, {pre}_mod, {len(rel.modulus)}, {pre}_pinv, {pre}_w, {wb}, {pre}_woff, {pre}_modv}},')
        out.append("};")
        out.append(f"constexpr int {cu}_n = {len(rels)};")
        out.append("")
    out.append("}}  // namespace")
    return "\n".join(out) + "\n"


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
