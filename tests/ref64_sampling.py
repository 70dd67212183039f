"""Independent float64 statement of the D3D sampling rules the path relies on — the second opinion for oracle/vqo_sampling.h, which is also
what serves the texture fetches of the reference's HLSL in oracle/ref_src/ref_hooks.cpp (so the reference-source fixtures cannot pin it).
Written from the D3D11.3 functional specification, not from the oracle:
  * texel-space coordinate x = u*N - 0.5; fixed point with 8 fractional bits (§7.18.7): fx = floor(x*256 + 0.5), texel = fx >> 8, weight = (fx & 255)/256;
  * bilinear = (1-wx)(1-wy) c00 + wx(1-wy) c10 + (1-wx)wy c01 + wx wy c11 in exact arithmetic;
  * trilinear = (1-f) lo + f hi, f = the 8-bit fraction of the (clamped) LOD; LOD of Sample() = log2 of the longer screen-space derivative;
  * address modes WRAP (modulo) and CLAMP;
  * cube maps (§7.18.11 / seamless filtering): face = major axis (ties z > y > x), (sc, tc)/|ma| -> [0,1]^2; a tap outside the face is the texel
    SEEN from the cube centre through the centre of that out-of-range texel of the extended face plane (ray cast onto the neighbouring
    face) — a geometric definition that does not share the oracle's integer edge table; a tap outside through a CORNER has no texel: it
    takes the mean of the other three taps.
Everything here is float64 / Python integers. The comparison (tests/test_sampler_ref64.py) allows for ONE 8-bit fraction step where the
oracle's binary32 coordinate lands on the other side of a 1/512 boundary."""
import numpy as np

# face bases as the reference's cube cameras build them (CubemapUtility.cpp:40-49, LookAtLH): forward F, up U, right R = U x F
FACES = [((1, 0, 0), (0, 1, 0)), ((-1, 0, 0), (0, 1, 0)), ((0, 1, 0), (0, 0, -1)), ((0, -1, 0), (0, 0, 1)), ((0, 0, 1), (0, 1, 0)), ((0, 0, -1), (0, 1, 0))]


def _basis(f):
    F, U = np.array(FACES[f][0], np.float64), np.array(FACES[f][1], np.float64)
    return F, U, np.cross(U, F)


def fixed8(x):
    fx = int(np.floor(np.float64(x) * 256.0 + 0.5))
    return fx >> 8, (fx & 255) / 256.0


def face_uv(d):
    """direction -> (face, su, sv) with sv growing DOWN the texture"""
    d = np.asarray(d, np.float64)
    a = np.abs(d)
    if a[2] >= a[0] and a[2] >= a[1]:
        f = 5 if d[2] < 0 else 4
    elif a[1] >= a[0]:
        f = 3 if d[1] < 0 else 2
    else:
        f = 1 if d[0] < 0 else 0
    F, U, R = _basis(f)
    ma = abs(float(d @ F))
    return f, (float(d @ R) / ma) * 0.5 + 0.5, (-(float(d @ U)) / ma) * 0.5 + 0.5


def texel_seen_through(f, i, j, N):
    """(face, i, j) of the texel hit by the ray from the centre through the centre of texel (i, j) of face f's plane; (i, j) may lie one
    texel outside [0, N). None for a corner (outside in both directions)."""
    ox, oy = i < 0 or i >= N, j < 0 or j >= N
    if ox and oy:
        return None
    if not ox and not oy:
        return f, i, j
    F, U, R = _basis(f)
    p = F + ((2 * i + 1) / N - 1.0) * R + (1.0 - (2 * j + 1) / N) * U
    g, su, sv = face_uv(p)
    return g, min(max(int(np.floor(su * N)), 0), N - 1), min(max(int(np.floor(sv * N)), 0), N - 1)


def sample_cube(cube, d):
    """cube: float64 [6,N,N,C]; returns (value, taps) — taps = the four tap colours (for the one-step bound)"""
    N = cube.shape[1]
    f, su, sv = face_uv(d)
    ix, wx = fixed8(su * N - 0.5)
    iy, wy = fixed8(sv * N - 0.5)
    taps = []
    for t in range(4):
        r = texel_seen_through(f, ix + (t & 1), iy + (t >> 1), N)
        taps.append(None if r is None else cube[r[0], r[2], r[1]].astype(np.float64))
    for t in range(4):
        if taps[t] is None:
            taps[t] = sum(taps[k] for k in range(4) if k != t and taps[k] is not None) / 3.0
    w = [(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy]
    return sum(w[t] * taps[t] for t in range(4)), taps


def sample_2d(tex, u, v, mode):
    """tex: [H,W,C]; mode 'wrap' | 'clamp'. Returns (value, taps)."""
    H, W = tex.shape[:2]
    ix, wx = fixed8(np.float64(u) * W - 0.5)
    iy, wy = fixed8(np.float64(v) * H - 0.5)

    def addr(i, n):
        return i % n if mode == "wrap" else min(max(i, 0), n - 1)
    taps = [tex[addr(iy + (t >> 1), H), addr(ix + (t & 1), W)].astype(np.float64) for t in range(4)]
    w = [(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy]
    return sum(w[t] * taps[t] for t in range(4)), taps


def split_lod(lod, n_mips):
    lod = 0.0 if not (lod > 0.0) else min(float(lod), float(n_mips - 1))        # NaN / -inf / negative -> 0
    fl = int(np.floor(lod * 256.0 + 0.5))
    lo, f = fl >> 8, (fl & 255) / 256.0
    if lo >= n_mips - 1:
        lo, f = n_mips - 1, 0.0
    return lo, f


def sample_chain(levels, u, v, lod, mode="wrap"):
    """levels: list of [h,w,C] arrays (level 0 first); explicit LOD, trilinear. Returns (value, taps of both levels)."""
    lo, f = split_lod(lod, len(levels))
    a, ta = sample_2d(levels[lo], u, v, mode)
    if f == 0.0:
        return a, ta
    b, tb = sample_2d(levels[lo + 1], u, v, mode)
    return (1 - f) * a + f * b, ta + tb


def sample_cube_chain(cubes, d, lod):
    """cubes: list of float64 [6,N>>m,N>>m,C] (mip 0 first); explicit LOD on a MIN_MAG_MIP_LINEAR sampler: trilinear between two seamless bilinear
    fetches. Returns (value, taps of both levels)."""
    lo, f = split_lod(lod, len(cubes))
    a, ta = sample_cube(cubes[lo], d)
    if f == 0.0:
        return a, ta
    b, tb = sample_cube(cubes[lo + 1], d)
    return (1 - f) * a + f * b, ta + tb


def lod_from_derivatives(ddx, ddy, w, h, bias=0.0):
    """Sample(): LOD = log2 of the longer of the two screen-space derivative vectors in texel units (isotropic), + bias"""
    dx = np.hypot(np.float64(ddx[0]) * w, np.float64(ddx[1]) * h)
    dy = np.hypot(np.float64(ddy[0]) * w, np.float64(ddy[1]) * h)
    m = max(dx, dy)
    return (-np.inf if m == 0 else np.log2(m)) + bias


def point_wrap(tex, u, v):
    """POINT_WRAP fetch with the coordinate snapped to 8 fractional bits first (§7.18.7)"""
    H, W = tex.shape[:2]
    tx = (int(np.floor(np.float64(u) * W * 256.0 + 0.5)) >> 8) % W
    ty = (int(np.floor(np.float64(v) * H * 256.0 + 0.5)) >> 8) % H
    return tex[ty, tx]
