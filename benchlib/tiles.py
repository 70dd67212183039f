"""`tile_curve`: per-tile step time of the cfg5 frame on ONE GPU and a labelled model of the speed-up (never quoted as a result)."""
from .consts import F16, R8, XGMI_LINK_GBPS
from .timing import _time_loop


def tile_curve(ctx, d, comms, args, p5, strong):
    """One GPU can say how the cfg5 tile step shrinks with the tile: step time for the top 4320/N rows, N = 1, 2, 4, 8 (same frame, same lights).
    The speed-up line is a MODEL, labelled as such: tile step + halo bytes / link + (composite bytes into the root over its N-1 links when it
    is not overlapped). It bounds compute efficiency and tail effects; RCCL latency and launch skew are not in it."""
    W, H = p5.W, p5.frame_h
    res = {"frame": [W, H], "lights": p5.cfg["lights"], "tiles": [], "t1_ms": None,
           "model": "speedup(N) = t_step(4320 rows) / (t_step(4320/N rows) + halo + composite_if_not_overlapped); halo = 2 x 10 rows x 7680 px x 8 B over one link, "
                    "composite = (N-1)/N x 132.7 MB into the root over N-1 links; link = 153 GB/s peak (76 GB/s in the conservative column). MODELLED, not measured."}
    t1 = None
    for n in (1, 2, 4, 8):
        rows = H // n
        gb = [g[:rows] for g in p5.gb]
        sc, xb, sd = p5.scene[0][:rows], p5.xblur[:rows], p5.sdr[0][:rows]

        def step(i):
            ctx.forward_lighting(gb, p5.pf, p5.pv, out=sc, out_fmt=F16, extra_point=p5.extra)
            ctx.gaussian_blur_x(sc, F16, out=xb)
            ctx.gaussian_blur_y_tonemap(xb, F16, R8, out=sd)
        ms = _time_loop(step, 8 * n if n < 8 else 40, 3 * n)
        t1 = t1 or ms
        halo_b, comp_b = 2 * 10 * W * 8, (n - 1) / n * W * H * 4
        row = {"gpus_modelled": n, "tile_rows": rows, "workgroups_shade": ((W + 255) // 256) * rows, "step_ms": round(ms, 4), "compute_speedup": round(t1 / ms, 3)}
        for tag, bw in (("peak_link", XGMI_LINK_GBPS), ("half_link", XGMI_LINK_GBPS / 2)):
            halo_ms = 0.0 if n == 1 else halo_b / bw / 1e6
            comp_ms = 0.0 if n == 1 else comp_b / (bw * (n - 1)) / 1e6
            row[f"modelled_speedup_overlapped_{tag}"] = round(t1 / (ms + halo_ms), 3)
            row[f"modelled_speedup_serial_{tag}"] = round(t1 / (ms + halo_ms + comp_ms), 3)
        res["tiles"].append(row)
    res["t1_ms"] = round(t1, 4)
    return res
