// dev_projective.hpp -- projective matcher (k_projective).
// Part of icp_device.hpp (included from there, inside namespace icpdev); see that file for the build contract.
// ------------------------------------------------------------------------------------------------
// Projective matcher, NearestNeighbor.h:333-421.  One lane = one query; the 25x25 window of the
// organised target is read through L1/L2 (neighbouring lanes' windows overlap almost entirely).
struct ProjParams {
    const float* sx; const float* sy; const float* sz; const int* sel; int n;
    const float* tx; const float* ty; const float* tz; int width; int height;
    float fx, fy, mx, my; int window;
    const PoseState* ps; int pretransformed; float max_dist;
    icp_match_t* out; float* d2_out;
};

__global__ __launch_bounds__(256) void k_projective(const ProjParams pp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= pp.n) return;
    const int i = pp.sel ? pp.sel[k] : k;
    float px = pp.sx[i], py = pp.sy[i], pz = pp.sz[i];
    if (!pp.pretransformed) { float a, b, c; xform_point(pp.ps->pose, px, py, pz, a, b, c); px = a; py = b; pz = c; }
    icp_match_t m; float best = FLT_MAX;
    if (px == -INFINITY) {                                   // :372-373 leaves the value-initialised Match{0, 0.f}
        m.idx = 0; m.weight = 0.f;
    } else {
        const float uf = roundf(((px * pp.fx) / pz) + pp.mx);    // :378
        const float vf = roundf(((py * pp.fy) / pz) + pp.my);    // :379
        const float wf = (float)pp.window;
        int bi = -1;
        // unsigned underflow (:385-386): a window starting below 0 never runs; NaN / negative / huge => no match
        if (uf >= wf && vf >= wf && uf < 2147483648.f && vf < 2147483648.f) {
            const long long u0 = (long long)uf - pp.window, u1 = (long long)uf + pp.window;
            const long long v0 = (long long)vf - pp.window, v1 = (long long)vf + pp.window;
            const int ve = (int)(v1 < (long long)pp.height - 1 ? v1 : (long long)pp.height - 1);
            const int ue = (int)(u1 < (long long)pp.width - 1 ? u1 : (long long)pp.width - 1);
            if (v0 < pp.height && u0 < pp.width) {
                // Two window columns per step (packed f32, 8-byte loads), row-major order and strict < kept (:399).  A target
                // hole (x == MINF, :392) needs no test: px - (-inf) = +inf makes its distance +inf (or NaN), never < best.
                const f2 px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz};
                for (int v = (int)v0; v <= ve; v++) {
                    const int row = v * pp.width;
                    int u = (int)u0;
                    for (; u + 3 <= ue; u += 4) {                         // four columns: 16-byte loads
                        const int j = row + u;
                        f2 qx[2], qy[2], qz[2];
                        __builtin_memcpy(qx, pp.tx + j, 16); __builtin_memcpy(qy, pp.ty + j, 16); __builtin_memcpy(qz, pp.tz + j, 16);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const f2 dx = px2 - qx[h], dy = py2 - qy[h], dz = pz2 - qz[h];
                            const f2 d = dx * dx + (dy * dy + dz * dz);
                            if (d.x < best) { best = d.x; bi = j + 2 * h; }
                            if (d.y < best) { best = d.y; bi = j + 2 * h + 1; }
                        }
                    }
                    for (; u + 1 <= ue; u += 2) {
                        const int j = row + u;
                        f2 qx, qy, qz;
                        __builtin_memcpy(&qx, pp.tx + j, 8); __builtin_memcpy(&qy, pp.ty + j, 8); __builtin_memcpy(&qz, pp.tz + j, 8);
                        const f2 dx = px2 - qx, dy = py2 - qy, dz = pz2 - qz;
                        const f2 d = dx * dx + (dy * dy + dz * dz);        // :396 Eigen squaredNorm tree
                        if (d.x < best) { best = d.x; bi = j; }
                        if (d.y < best) { best = d.y; bi = j + 1; }
                    }
                    if (u <= ue) {
                        const int j = row + u;
                        const float dx = px - pp.tx[j], dy = py - pp.ty[j], dz = pz - pp.tz[j];
                        const float d = dx * dx + (dy * dy + dz * dz);
                        if (d < best) { best = d; bi = j; }
                    }
                }
            }
        }
        if (best <= pp.max_dist) { m.idx = bi; m.weight = 1.f; } else { m.idx = -1; m.weight = 0.f; }   // :407-415
    }
    pp.out[k] = m;
    if (pp.d2_out) pp.d2_out[k] = best;
}
