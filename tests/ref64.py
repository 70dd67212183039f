"""Second-opinion float64 numpy restatement of the closed-form parts of the path (no texture sampling), written
independently of oracle/*.cpp straight from the HLSL formulas. Used by the CPU tests to check the ORACLE's logic
(tolerance-level, not bits): a transcription slip in the oracle shows up as a gross mismatch here.
Cites: Shaders/BRDF.hlsl:65-194, Shaders/Lighting.hlsl:29-32,57-73,308-345, Shaders/ForwardLighting.hlsl:284-293,
Shaders/Tonemapper.hlsl:24-27, Shaders/HDR.hlsl:76-119, Shaders/GaussianBlur.hlsl:109-148."""
import numpy as np

PI = 3.14159265359
W21 = np.array([0.224716, 0.191756, 0.119146, 0.053897, 0.017746, 0.004252, 0.000741, 0.000094, 0.000009, 0.000001, 0.0])


def _n(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def _dot(a, b):
    return (a * b).sum(-1)


def brdf(N, rough, albedo, metal, Wi, V):
    Wo, N = _n(V), _n(N)
    H = _n(Wo + Wi)
    NdotH, NdotV, NdotL = (np.clip(_dot(N, x), 0, 1) for x in (H, Wo, Wi))
    F0 = 0.04 + metal[..., None] * (albedo - 0.04)
    F = F0 + (1 - F0) * ((1 - np.maximum(0, _dot(H, V))) ** 5)[..., None]
    k = (rough + 1) ** 2 / 8

    def g1(X):
        nv = np.maximum(0, _dot(N, X))
        return nv / (nv * (1 - k) + k + 0.0001)
    G = g1(Wo) * g1(Wi)
    a2 = rough ** 4
    den = PI * (NdotH ** 2 * (a2 - 1) + 1) ** 2
    D = np.where(den < 1e-12, 1.0, a2 / np.maximum(den, 1e-300))
    spec = (D * G / np.maximum(4 * NdotV * NdotL, 0.0001))[..., None] * F
    kD = (1 - F) * (1 - metal)[..., None]
    return kD * albedo / PI + spec


def shade(gb, cam, points=(), spots=(), directional=None):
    """gb: 4 arrays [...,4] float; lights: lists of dicts. No IBL, no shadows."""
    g0, g1, g2, g3 = (np.asarray(g, np.float64) for g in gb)
    P, ao = g0[..., :3], g0[..., 3]
    N, rough = g1[..., :3], g1[..., 3]
    alb, metal = g2[..., :3], g2[..., 3]
    V = _n(np.asarray(cam, np.float64) - P)
    I = alb * ao[..., None] + g3[..., :3] * g3[..., 3:4]
    for l in points:
        d = np.asarray(l["pos"], np.float64) - P
        D = np.sqrt(_dot(d, d))
        Wi = d / D[..., None]
        rad = (1 / (D * D))[..., None] * np.asarray(l["color"]) * l["brightness"]
        c = brdf(N, rough, alb, metal, Wi, V) * rad * np.clip(_dot(N, Wi), 0, 1)[..., None]
        I = I + np.where((D < l["range"])[..., None], c, 0)
    for l in spots:
        d = np.asarray(l["pos"], np.float64) - P
        D = np.sqrt(_dot(d, d))
        Wi = d / D[..., None]
        theta = np.arccos(np.clip(_dot(_n(-d), _n(np.asarray(l["dir"], np.float64))), -1, 1))
        cone = np.where(theta > l["outer"], 0.0, np.where(theta <= l["inner"], 1.0, 1 - (theta - l["inner"]) / (l["outer"] - l["inner"])))
        rad = (cone / (D * D))[..., None] * np.asarray(l["color"]) * l["brightness"]
        I = I + brdf(N, rough, alb, metal, Wi, V) * rad * np.clip(_dot(N, Wi), 0, 1)[..., None]
    if directional is not None:
        Wi = np.broadcast_to(_n(-np.asarray(directional["dir"], np.float64)), P.shape)
        rad = np.asarray(directional["color"]) * directional["brightness"]
        I = I + brdf(N, rough, alb, metal, Wi, V) * rad * np.clip(_dot(N, Wi), 0, 1)[..., None]
    return np.concatenate([I, rough[..., None]], -1)


def blur1d(img, axis):
    img = np.asarray(img, np.float64)
    n = img.shape[axis]
    out = np.zeros_like(img)
    for k in range(-10, 11):
        idx = np.clip(np.arange(n) + k, 0, n - 1)
        out += np.take(img, idx, axis=axis) * W21[abs(k)]
    out[..., 3] = 1.0
    return out


def tonemap_srgb(img, gamma=True):
    c = np.asarray(img, np.float64)[..., :3]
    t = c / (c + 1)
    if gamma:
        t = np.where(t < 0.0031308, 12.92 * t, 1.055 * np.abs(t) ** (1 / 2.4) - 0.055)
    return t


def tonemap_pq(img, nits=200.0, rec709=True):
    c = np.asarray(img, np.float64)[..., :3]
    if rec709:
        M = np.array([[0.627402, 0.329292, 0.043306], [0.069095, 0.919544, 0.011360], [0.016394, 0.088028, 0.895578]])
        c = c @ M.T
    c = np.abs(c * (nits / 10000.0))
    m1, m2, c1, c2, c3 = 2610 / 4096 / 4, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
    cp = c ** m1
    return ((c1 + c2 * cp) / (1 + c3 * cp)) ** m2


def radical_inverse(i):
    i = np.asarray(i, np.uint64)
    r = np.zeros(i.shape, np.float64)
    for b in range(32):
        r += ((i >> np.uint64(b)) & np.uint64(1)) * 2.0 ** -(b + 1)
    return r


def integrate_brdf(ndotv, rough, count):
    """BRDF.hlsl:IntegrateBRDF with N = (0,0,1) in float64."""
    i = np.arange(count)
    xi_x, xi_y = i / count, radical_inverse(i)
    a = rough * rough
    phi = 2 * PI * xi_x
    ct = np.sqrt((1 - xi_y) / (1 + (a * a - 1) * xi_y))
    st = np.sqrt(1 - ct * ct)
    hx, hy = np.cos(phi) * st, np.sin(phi) * st
    H = np.stack([hy, -hx, ct], -1)            # tangent (0,-1,0), bitangent (1,0,0), N (0,0,1)
    V = np.array([np.sqrt(1 - ndotv * ndotv), 0, ndotv])
    L = _n(-V + 2 * _dot(V, H)[:, None] * H)
    nl, nh, vh = np.maximum(L[:, 2], 0), np.maximum(H[:, 2], 0), np.maximum(_dot(V, H), 0)
    k = rough * rough / 2

    def g1(c):
        c = np.maximum(0, c)
        return c / (c * (1 - k) + k + 0.0001)
    G = g1(ndotv) * g1(L[:, 2])
    m = nl > 0
    gv = np.maximum(G[m] * vh[m] / (nh[m] * ndotv), 0.0001)
    fc = (1 - vh[m]) ** 5
    return ((1 - fc) * gv).sum() / count, (fc * gv).sum() / count
