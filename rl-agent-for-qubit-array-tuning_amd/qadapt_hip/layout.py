"""Python mirror of csrc/qd_common.h::qd_layout -- offsets (in float64 elements)
of the per-env parameter and state blocks.  tests/test_layout.py checks it
against the compiled library (qd_layout_query)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class Layout:
    N: int
    G: int
    nb: int
    V: int
    cdd_inv: int
    cgd: int
    cbg: int
    ufac: int
    uinv: int
    alpha: int
    origin: int
    vopt: int
    vbopt: int
    pmin: int
    pmax: int
    bmin: int
    bmax: int
    scal: int
    noise: int
    pleads: int
    pinter: int
    size: int
    s_vgm: int
    s_gate_v: int
    s_barrier_v: int
    s_gate_gt: int
    s_barrier_gt: int
    s_sensor_gt: int
    s_kmean: int
    s_kvar: int
    s_size: int


def layout(N: int) -> Layout:
    G, nb, V = N + 1, N - 1, 2 * N
    o = 0
    f = {}
    for name, n in (("cdd_inv", G * G), ("cgd", G * V), ("cbg", nb * G), ("ufac", N * N), ("uinv", N),
                    ("alpha", nb), ("origin", G), ("vopt", G), ("vbopt", nb), ("pmin", N),
                    ("pmax", N), ("bmin", nb), ("bmax", nb), ("scal", 8), ("noise", 8), ("pleads", N), ("pinter", N * N)):
        f[name] = o
        o += n
    f["size"] = (o + 1) & ~1
    o = 0
    for name, n in (("s_vgm", G * G), ("s_gate_v", N), ("s_barrier_v", nb), ("s_gate_gt", N),
                    ("s_barrier_gt", nb), ("s_sensor_gt", 1), ("s_kmean", N * N), ("s_kvar", N * N)):
        f[name] = o
        o += n
    f["s_size"] = (o + 1) & ~1
    return Layout(N=N, G=G, nb=nb, V=V, **f)


LAYOUT_FIELDS = ["N", "G", "nb", "V", "cdd_inv", "cgd", "cbg", "ufac", "uinv", "alpha", "origin", "vopt",
                 "vbopt", "pmin", "pmax", "bmin", "bmax", "scal", "noise", "pleads", "pinter", "size", "s_vgm", "s_gate_v",
                 "s_barrier_v", "s_gate_gt", "s_barrier_gt", "s_sensor_gt", "s_kmean", "s_kvar",
                 "s_size"]
