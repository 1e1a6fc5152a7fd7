// cc_k_segcells.h — the per-cell part of the ground segmentation (seg_pre_cells): shared by k_insert_par (fused front half) and k_seg_pre.
// (part of cc_kernels.h: included there, in order, inside namespace cck)
#pragma once

// =====================================================================================================
// ground-point segmentation — continuous_clustering.cpp:294-624, split into k_seg_pre (per cell) and k_seg_scan (per column)
// =====================================================================================================
__device__ __forceinline__ float len2(float a, float b)
{
    return ccm::sqrt_rn(a * a + b * b);
}

constexpr int EGO_STRIDE = 16; // doubles per firing in k_ego's output: {R 3x3, t, skip_r2, -}

// ---- the per-cell part of the segmentation of ONE column (everything of cc.cpp:306-403, 567-603 that does not depend on other columns or on
// the rows below), lanes = rows, cells in registers. Shared by k_seg_pre (cells from the ring) and k_insert_par (cells it has just computed).
//   x, y, z, dist, incl : the cell (odom frame; dist = incl = NaN without a return), inten its intensity
//   sp*                 : sgps_sensor_position of the column's job (the finishing firing's pose, cc.cpp:111-113, 291)
//   E                   : that firing's k_ego record (wave-uniform pointer: scalar loads)
// Staging for k_seg_scan: x2, uz (the point in the azimuth plane of the job's sensor position), flags (SG_*), and ONE more float w:
//   cell with a return, inclination step to the row below valid  w = that step (the column's own entry of the table, cc.cpp:353-357: k_seg_scan
//                                                                  takes the last valid one along the columns), cc.cpp:597-603 decided here
//   cell with a return, step not valid (SG_PENDING)              w = distance (k_seg_scan evaluates cc.cpp:597-603 once it knows the table)
//   cell without a return (SG_NAN)                               w = raw inclination of the row below (where the supplement chain of
//                                                                  cc.cpp:364-369 starts when that row has a return)
template<int RPL>
__device__ __forceinline__ void seg_pre_cells(const cc_config& cfg, const int R, const int lane, const float (&cx)[RPL], const float (&cy)[RPL],
                                              const float (&cz)[RPL], const float (&dist)[RPL], const float (&incl)[RPL], const uint8_t (&inten)[RPL],
                                              const float spx, const float spy, const float spz, const double* __restrict__ E, float (&x2)[RPL],
                                              float (&uz)[RPL], float (&w)[RPL], int (&flags)[RPL])
{
    // raw inclination of the row below (0 below the last row, cc.cpp:312)
    float below[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const float nxt0 = (k + 1 < RPL) ? __shfl(incl[(k + 1 < RPL) ? k + 1 : k], 0, 64) : 0.f;
        const float dn = __shfl_down(incl[k], 1, 64);
        below[k] = lane == 63 ? nxt0 : dn;
        if (k * 64 + lane + 1 >= R)
            below[k] = 0.f;
    }
    const float skip_r2 = (float) E[12];
    bool close = false, need_exact = false;
    bool incl_ignore[RPL];
#pragma unroll
    for (int k = 0; k < RPL; k++)
    {
        const int row = k * 64 + lane;
        flags[k] = SG_NAN;
        x2[k] = uz[k] = 0.f;
        w[k] = below[k];
        incl_ignore[k] = false;
        if (row >= R)
            continue;
        const bool isnan_ = dist[k] != dist[k];
        if (isnan_)
            continue;
        int f = 0;
        if (cfg.fog_filtering_enabled && inten[k] < (uint8_t) cfg.fog_filtering_intensity_below && dist[k] < cfg.fog_filtering_distance_below &&
            incl[k] > cfg.fog_filtering_inclination_above)
            f |= SG_FOG;
        const float ux = cx[k] - spx, uy = cy[k] - spy;
        uz[k] = cz[k] - spz;
        x2[k] = len2(ux, uy);
        const float r2 = x2[k] * x2[k] + uz[k] * uz[k];
        if (!(r2 > skip_r2))
        {
            f |= SG_EGO; // provisional: "needs the transform"
            close = true;
        }
        if (dist[k] < cfg.max_distance) // (cc.cpp:590: distance < 1. * max_distance in double — both convert exactly, the same comparison)
            f |= SG_TOO_CLOSE;
        const float diff = incl[k] - below[k];
        if (diff != diff)
        {
            f |= SG_PENDING;
            w[k] = dist[k];
        }
        else
        {
            w[k] = diff;
            // cc.cpp:597-603: atan2f(max_distance, distance) < inclination step to the next laser. The exact (glibc-identical) atan2f
            // costs ~100 instructions per wave, and the test can only hold beyond ~100 m: a rigorous filter first. With
            // x = max_distance / distance >= 1.01 t (0 <= t < 0.05): atan(x) >= x - x^3/3 >= 1.006 t for x <= 0.1, atan(x) > 0.099 > t
            // otherwise, and atan2f is within an ulp of atan — so the test is false without evaluating it.
            if (cfg.ignore_points_with_too_big_inclination_angle_diff && row < (R - 1))
            {
                const bool surely_false = cfg.max_distance > 0.f && diff >= 0.f && diff < 0.05f && cfg.max_distance >= 1.01f * dist[k] * diff;
                incl_ignore[k] = !surely_false; // provisional: "needs the exact evaluation"
                need_exact |= !surely_false;
            }
        }
        flags[k] = f;
    }
    if (__any(need_exact))
    {
#pragma unroll
        for (int k = 0; k < RPL; k++)
            if (incl_ignore[k])
                incl_ignore[k] = ccm::atan2f_exact(cfg.max_distance, dist[k]) < w[k];
    }
#pragma unroll
    for (int k = 0; k < RPL; k++)
        if (incl_ignore[k])
            flags[k] |= SG_INCL_IGNORE;
    if (__any(close))
    {
        // ego_robot_frame_from_odom_frame * point (cc.cpp:390-403), Eigen's evaluation order
        double er[9], et[3];
        for (int i = 0; i < 9; i++)
            er[i] = E[i];
        for (int i = 0; i < 3; i++)
            et[i] = E[9 + i];
#pragma unroll
        for (int k = 0; k < RPL; k++)
        {
            if (!(flags[k] & SG_EGO) || (flags[k] & SG_NAN))
                continue;
            const double dx = cx[k], dy = cy[k], dz = cz[k];
            const double ex = ((er[0] * dx + er[1] * dy) + er[2] * dz) + et[0];
            const double ey = ((er[3] * dx + er[4] * dy) + er[5] * dz) + et[1];
            const double ez = ((er[6] * dx + er[7] * dy) + er[8] * dz) + et[2];
            const bool in_box = ex < cfg.length_ref_to_front_end_ && ex > cfg.length_ref_to_rear_end_ && ey < cfg.width_ref_to_left_mirror_ &&
                                ey > cfg.width_ref_to_right_mirror_ && ez < cfg.height_ref_to_maximum_ && ez > cfg.height_ref_to_ground_;
            if (!in_box)
                flags[k] &= ~SG_EGO;
        }
    }
}
