"""CPU study for the fused strict trace (round 5): how soon does the Newton iterate of a ray become PERIODIC?

The reference iterates every ray of a call the batch-wide number of times (`while (|ft| > 5e-5).any()`, deeplens/surfaces.py:547).
t_{j+1} = f(t_j) with o, d, mask fixed is a deterministic float32 map, so once t_j repeats (period 1: t_j == t_{j-1}; period 2:
t_j == t_{j-2}) every later iterate is known without evaluating it.  This script runs the ORACLE's loop (oracle/lens.py, the pinned
restatement) on the bench stack's first slice and reports, per curved surface: the batch-wide count, the distribution of the
iteration at which each ray's cycle is detected, and the same maximum over groups of 64 rays (a wavefront) for two ray orders.
Oracle only - runs on the CPU."""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch

from oracle import lens as OL

rec = []


def newton(self, o, d, ra):
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    t0 = (self.d - o[..., 2]) / dz
    t = t0
    ft = OL.MAXT * torch.ones_like(o[..., 2])
    it = 0

    def residual(t, mask_fn):
        p = o + d * t.unsqueeze(-1)
        px, py = p[..., 0], p[..., 1]
        m = mask_fn(px, py) & (ra > 0)
        xm, ym = px * m, py * m
        ft = self.sag_r2(xm ** 2 + ym ** 2) + self.d - p[..., 2]
        dr2dt = 2 * ((dx ** 2 + dy ** 2) * t + (dx * o[..., 0] + dy * o[..., 1]))
        dfdt = self.dsag_dr2(xm ** 2 + ym ** 2) * dr2dt - dz
        return ft, dfdt

    ts = [t.clone()]
    above = []
    while (torch.abs(ft) > OL.TOL_LOOSE).any() and it < OL.NEWTON_MAXITER:
        it += 1
        ft, dfdt = residual(t, self.valid_loose)
        above.append((torch.abs(ft) > OL.TOL_LOOSE))
        t = t - torch.clamp(ft / (dfdt + OL.EPSILON), -OL.STEP_BOUND, OL.STEP_BOUND)
        ts.append(t.clone())
    rec.append((it, torch.stack(ts).view(torch.int32).numpy().reshape(len(ts), -1), torch.stack(above).numpy().reshape(it, -1),
                (ra > 0).numpy().reshape(-1)))
    t1 = t - t0
    t = t0 + t1
    ft, dfdt = residual(t, self.valid_strict)
    t = t - torch.clamp(ft / (dfdt + OL.EPSILON), -OL.STEP_BOUND, OL.STEP_BOUND)
    p = o + d * t.unsqueeze(-1)
    valid = self.valid_strict(p[..., 0], p[..., 1]) & (torch.abs(ft) < OL.TOL_TIGHT) & (ra > 0) & (t > 0)
    return valid, t


OL.Surf.newton = newton


def detect(ts, n):
    """iteration (1-based count of residual evaluations) after which the cycle of each ray is known; n if never."""
    done = np.full(ts.shape[1], n, dtype=np.int32)
    for j in range(1, n + 1):
        hit = ts[j] == ts[j - 1]
        if j >= 2:
            hit |= ts[j] == ts[j - 2]
        done = np.where((done == n) & hit, j, done)
    return done


def main():
    grid, spp = 11, int(os.environ.get("SPP", 256))
    lens = OL.OracleLens(os.path.join(ROOT, "lenses/rf50mm/lens.json"), sensor_res=(1024, 1024))
    torch.manual_seed(0)
    for which, focus in (("near focus", -500.0), ("far focus", -5000.0)):
        lens.refocus(focus)
        pts = lens.point_source_grid(-3167.0, grid).reshape(-1, 3)
        pobj = lens.object_points(pts)
        for name, shrink in (("main", False), ("chief", True)):
            rec.clear()
            lens.trace2sensor(lens.sample_from_points(pobj, spp=spp, shrink_pupil=shrink))
            N = grid * grid
            print(f"--- {which}, {name} batch: {spp} x {N} rays")
            tot_now = tot_cyc = tot_wave_a = tot_wave_b = tot_alive = tot_dead = 0
            for si, (n, ts, above, alive) in enumerate([r for r in rec if r[1].shape[1] == spp * N]):
                done = detect(ts, n)
                # wave = 64 consecutive rays of the flat order (sample-major), or 64 samples of one point
                wa = done[: len(done) // 64 * 64].reshape(-1, 64).max(1)
                wb = done.reshape(spp, N).T.reshape(-1)[: len(done) // 64 * 64].reshape(-1, 64).max(1)
                never = (done == n) & ~((ts[n] == ts[n - 1]) | (ts[n] == ts[max(n - 2, 0)]))
                last_above = above[n - 1]
                print(f"  curved surface {si}: batch count {n:2d}; alive {alive.mean():.3f}; per-ray evaluations mean {done.mean():.2f} "
                      f"(p50 {np.percentile(done, 50):.0f} p99 {np.percentile(done, 99):.0f} max {done.max()}); "
                      f"per-wave max: flat order {wa.mean():.2f}, per-point order {wb.mean():.2f}; no cycle by the end {never.mean():.4f}; "
                      f"rays above tol in the last iteration {last_above.sum()} (alive among them {int((last_above & alive).sum())})")
                da = np.where(alive, done, 0)
                wal = da[: len(done) // 64 * 64].reshape(-1, 64).max(1)
                dead = done[~alive]
                print(f"      alive rays only: mean {done[alive].mean():.2f}, max {done[alive].max()}, per-wave max (dead lanes idle) {wal.mean():.2f}; "
                      f"dead rays: {len(dead)} mean {dead.mean() if len(dead) else 0:.2f} p90 {np.percentile(dead, 90) if len(dead) else 0:.0f}")
                tot_alive += wal.mean(); tot_dead += dead.sum() / len(done)
                tot_now += n; tot_cyc += done.mean(); tot_wave_a += wa.mean(); tot_wave_b += wb.mean()
            print(f"  residual evaluations per ray over the curved surfaces: batch-wide {tot_now}, per-ray cycle {tot_cyc:.1f}, "
                  f"per wave (flat) {tot_wave_a:.1f}, per wave (per point) {tot_wave_b:.1f}; alive lanes only per wave {tot_alive:.1f} + dead rays in dense waves {tot_dead:.1f}")


if __name__ == "__main__":
    main()
