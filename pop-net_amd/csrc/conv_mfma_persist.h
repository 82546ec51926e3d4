// Persistent variant of the MFMA convolution (same tiling, fragments, LDS image and arithmetic as
// conv_mfma_kernel in conv_mfma_kernel.h; bit-identical results).
//
// Why: at batch 32 a layer is only a few hundred tiles of ~1-7 us of matrix work each, so the
// per-block fixed costs -- first halo fetch (HBM/MALL round trip), weight-queue priming, the
// epilogue's residual loads and stores, and the tail of a grid that is 1.3 "waves" of blocks --
// dominated the v1 launch (profiles/r01_*).  Here a launch is exactly 2 blocks per CU; each block
// pulls tiles from one atomic queue (longest problems first: near-ideal balance without stream-K
// style splitting) and, BEFORE a tile's epilogue, fetches the next tile's first halo chunk into
// registers and primes its weight queue, so those latencies overlap the LDS transpose + stores.
//
// Inter-workgroup communication: only the relaxed atomic tile counter (zeroed by a memset node at
// the start of every forward); no data is handed between workgroups inside a launch.
#pragma once
#include "conv_mfma_kernel.h"

struct PnTile {                 // wave-uniform description of one output tile
    const ConvProblem *P;
    int cb, b, oy0, ox0, Wc, npix, HC, npx, iy0, ix0;
    float inv_wc, inv_hc;
};

template <int KS, int STRIDE>
__device__ __forceinline__ PnTile pn_make_tile(const ConvProblem *__restrict__ probs, int nprob, int t) {
    int pi = 0;
    for (int i = 1; i < nprob; ++i)
        if (t >= probs[i].tile_base) pi = i;        // tile_base ascending, problems listed longest first
    PnTile T;
    const ConvProblem *P = probs + pi;
    T.P = P;
    const int bx = t - P->tile_base;
    T.cb = bx % P->cout_blocks;
    const int tt = bx / P->cout_blocks;
    const int tile = tt % P->tiles_per_img;
    T.b = tt / P->tiles_per_img;
    const int ty = tile / P->tiles_x, tx = tile - ty * P->tiles_x;
    T.oy0 = ty * P->R;
    T.ox0 = tx * P->Wt;
    const int R = min(P->R, P->Ho - T.oy0);
    T.Wc = min(P->Wt, P->Wo - T.ox0);
    T.npix = R * T.Wc;
    T.HC = (T.Wc - 1) * STRIDE + KS;
    T.npx = ((R - 1) * STRIDE + KS) * T.HC;
    T.iy0 = T.oy0 * STRIDE - KS / 2;
    T.ix0 = T.ox0 * STRIDE - KS / 2;
    T.inv_wc = 1.0f / (float)T.Wc;
    T.inv_hc = 1.0f / (float)T.HC;
    return T;
}

template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
__global__ __launch_bounds__(256, 2) void conv_mfma_persist_kernel(const ConvProblem *__restrict__ probs, int nprob, int total,
                                                                    int *__restrict__ counter, int next_off) {
    typedef Elem<PREC> E;
    typedef typename E::T T;
    typedef typename E::Frag Frag;
    constexpr int PIXB = E::PIXB, FRAGB = E::FRAGB, SUBX = E::SUBX;
    constexpr int WC = TileCfg<CFG>::WC, WP = TileCfg<CFG>::WP, CT = TileCfg<CFG>::CT, PT = TileCfg<CFG>::PT;
    constexpr int KK = KS * KS;
    constexpr int NCH = PIXB / 16, PPI = 256 / NCH;
    constexpr int NSTEP = KK * 2;
    constexpr int NA = (PREC == PN_PREC_BF16) ? ((NSTEP % 6 == 0) ? 6 : 2) : ((NSTEP % 3 == 0) ? 3 : 2);
    constexpr int NITEM = NSTEP * PT;
    constexpr int DB = (PREC == PN_PREC_BF16) ? ((NITEM % 3 == 0) ? 3 : 2) : 1;
    constexpr int MAXST = StageCfg<PREC, KS, STRIDE, PITCH, CFG>::MAXST;
    constexpr int BC = WC * CT * 16, ROWB = BC * 4 + 16, G = BC / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile int *s_next = reinterpret_cast<volatile int *>(smem + next_off);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int c = lane & 15, q = lane >> 4;
    const int ch = tid % NCH, p0 = tid / NCH;

    // ---- staging helpers (tile passed explicitly: the current and the next tile are both live) ----
    auto halo_src = [&](const PnTile &t, int p, int chunk, bool &inb) -> gcptr {
        int hy = (int)(((float)p + 0.5f) * t.inv_hc);
        int hx = p - hy * t.HC;
        int iy = t.iy0 + hy, ix = t.ix0 + hx;
        inb = p < t.npx && (unsigned)iy < (unsigned)t.P->H && (unsigned)ix < (unsigned)t.P->W;
        iy = min(max(iy, 0), t.P->H - 1);
        ix = min(max(ix, 0), t.P->W - 1);
        size_t pix = (size_t)(t.b * t.P->H + iy) * t.P->W + ix;
        return (gcptr)t.P->in + ((size_t)t.P->in_coff + pix * t.P->in_cs + (size_t)chunk * 64) * sizeof(T) + ch * 16;
    };
    auto halo_dst = [&](const PnTile &t, int p) -> int {
        int hy = (int)(((float)p + 0.5f) * t.inv_hc);
        int hx = p - hy * t.HC;
        int hp = hy * PITCH + hx;
        if (PREC == PN_PREC_BF16) return hp * PIXB + ((ch ^ (hp & 7)) << 4);
        return hp * PIXB + ((((ch >> 1) ^ (hp & 7)) << 5) | ((ch & 1) << 4));
    };
    u32x4 st[MAXST > 0 ? MAXST : 1];
    auto stage_load = [&](const PnTile &t, int chunk) {
#pragma unroll
        for (int it = 0; it < MAXST; ++it) {
            bool inb;
            st[it] = *reinterpret_cast<const PN_GLOBAL u32x4 *>(halo_src(t, p0 + it * PPI, chunk, inb));
        }
    };
    auto stage_store = [&](const PnTile &t, char *buf) {
#pragma unroll
        for (int it = 0; it < MAXST; ++it) {
            const int p = p0 + it * PPI;
            bool inb;
            (void)halo_src(t, p, 0, inb);
            if (p < t.npx) *reinterpret_cast<u32x4 *>(buf + halo_dst(t, p)) = inb ? st[it] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto stage_direct = [&](const PnTile &t, int chunk, char *buf) {
        for (int pb = p0; pb < t.npx; pb += 4 * PPI) {
            u32x4 v[4];
            bool inb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const PN_GLOBAL u32x4 *>(halo_src(t, pb + k * PPI, chunk, inb[k]));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (pb + k * PPI < t.npx) *reinterpret_cast<u32x4 *>(buf + halo_dst(t, pb + k * PPI)) = inb[k] ? v[k] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    int baddr[PT][KS];
    auto make_baddr = [&](const PnTile &t) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int slot = (wp * PT + pt) * 16 + c;
            int s = slot < t.npix ? slot : 0;
            int ry = (int)(((float)s + 0.5f) * t.inv_wc);
            int rx = s - ry * t.Wc;
            int hp0 = ry * STRIDE * PITCH + rx * STRIDE;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                int hp = hp0 + kx;
                baddr[pt][kx] = hp * PIXB + ((q ^ (hp & 7)) << (PREC == PN_PREC_BF16 ? 4 : 5));
            }
        }
    };
    gcptr wptr[CT];
    Frag aq[NA][CT];
    auto prime_weights = [&](const PnTile &t) {
        const int ctile0 = (t.cb * WC + wc) * CT;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wptr[ct] = (gcptr)t.P->wpack + (size_t)(ctile0 + ct) * t.P->ksteps * FRAGB + lane * 16;
#pragma unroll
        for (int d = 0; d < NA - 1; ++d)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) aq[d][ct] = load_a_frag<PREC>(wptr[ct] + d * FRAGB);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wptr[ct] += (NA - 1) * FRAGB;
    };

    int t_id = blockIdx.x;                       // first tile is static; grid <= total
    PnTile cur = pn_make_tile<KS, STRIDE>(probs, nprob, t_id);
    make_baddr(cur);
    prime_weights(cur);
    if (MAXST > 0) stage_load(cur, 0);

    for (;;) {
        int fetched = 0;
        if (tid == 0) fetched = (int)gridDim.x + atomicAdd(counter, 1);       // id of this block's next tile
        const ConvProblem &P = *cur.P;
        char *buf0 = smem, *buf1 = P.lds_two ? smem + P.lds_buf_bytes : smem;
        if (MAXST > 0) stage_store(cur, buf0);
        else stage_direct(cur, 0, buf0);
        f32x4 acc[CT][PT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();

        const int nchunks = P.cin_chunks;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const char *sm = (chunk & 1) ? buf1 : buf0;
            char *nbuf = (chunk & 1) ? buf0 : buf1;
            const bool more = chunk + 1 < nchunks;
            if (MAXST > 0 && more) stage_load(cur, chunk + 1);
#define PN_BADDR(j) ((baddr[(j) % PT][(((j) / PT) / 2) % KS] ^ ((((j) / PT) % 2) * SUBX)) + ((((j) / PT) / 2) / KS) * PITCH * PIXB)
            Frag bq[DB];
            if (DB > 1) {
#pragma unroll
                for (int j = 0; j < DB - 1; ++j) bq[j] = read_b_frag<PREC>(sm, PN_BADDR(j));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NITEM; ++j) {
                const int s = j / PT, pt = j % PT;
                if (pt == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        aq[(s + NA - 1) % NA][ct] = load_a_frag<PREC>(wptr[ct]);
                        wptr[ct] += FRAGB;
                    }
                }
                if (DB > 1) {
                    if (j + DB - 1 < NITEM) bq[(j + DB - 1) % DB] = read_b_frag<PREC>(sm, PN_BADDR(j + DB - 1));
                } else {
                    bq[0] = read_b_frag<PREC>(sm, PN_BADDR(j));
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct][pt] = mma(aq[s % NA][ct], bq[DB > 1 ? j % DB : 0], acc[ct][pt]);
                if (DB > 1) {
                    if (pt == 0) __builtin_amdgcn_sched_group_barrier(0x020, CT, 0);
                    if (j + DB - 1 < NITEM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                }
            }
#undef PN_BADDR
            if (more) {
                if (!P.lds_two) __syncthreads();
                if (MAXST > 0) stage_store(cur, nbuf);
                else stage_direct(cur, chunk + 1, nbuf);
                __syncthreads();
            }
        }

        // ---- hand-over: learn the next tile, start its memory traffic, then finish this tile ----
        if (tid == 0) *s_next = fetched;
        __syncthreads();                                 // also: every wave is done with the halo image
        const int nt = *s_next;
        const bool has_next = nt < total;
        PnTile nxt = cur;
        if (has_next) {
            nxt = pn_make_tile<KS, STRIDE>(probs, nprob, nt);
            prime_weights(nxt);
            if (MAXST > 0) stage_load(nxt, 0);
        }

        // epilogue of `cur` (identical arithmetic to conv_mfma_kernel)
        const int act = P.act;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = (wc * CT + ct) * 16 + 4 * q;
            const f32x4 bias4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cur.cb * BC + col);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int slot = (wp * PT + pt) * 16 + c;
                *reinterpret_cast<f32x4 *>(smem + slot * ROWB + col * 4) = acc[ct][pt] + bias4;
            }
        }
        __syncthreads();
        const int Wo = P.Wo;
        if (P.out) {
            for (int i = tid; i < cur.npix * G; i += 256) {
                const int slot = i / G, cg = i % G;
                const int co = cur.cb * BC + cg * 8;
                if (co >= P.cout) continue;
                const int ry = (int)(((float)slot + 0.5f) * cur.inv_wc);
                const int rx = cur.ox0 + (slot - ry * cur.Wc);
                const size_t opix = (size_t)(cur.b * P.Ho + cur.oy0 + ry) * Wo + rx;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(smem + slot * ROWB + cg * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(smem + slot * ROWB + cg * 32 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bool full = co + 7 < P.cout;
                if (P.res) {
                    const PN_GLOBAL T *rp = (const PN_GLOBAL T *)P.res + opix * P.res_cs + P.res_coff + co;
                    if (full) {
                        T rv[8];
                        if (sizeof(T) == 2) {
                            *reinterpret_cast<u32x4 *>(rv) = *reinterpret_cast<const PN_GLOBAL u32x4 *>(rp);
                        } else {
                            reinterpret_cast<u32x4 *>(rv)[0] = reinterpret_cast<const PN_GLOBAL u32x4 *>(rp)[0];
                            reinterpret_cast<u32x4 *>(rv)[1] = reinterpret_cast<const PN_GLOBAL u32x4 *>(rp)[1];
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
                    } else {
                        for (int k = 0; k < 8; ++k)
                            if (co + k < P.cout) v[k] += (float)rp[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = pn_activate(v[k], act, co + k, P.yolo_naf);
                PN_GLOBAL T *op = (PN_GLOBAL T *)P.out + opix * P.out_cs + P.out_coff + co;
                if (full) {
                    T ov[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) ov[k] = (T)v[k];
                    if (sizeof(T) == 2) {
                        *reinterpret_cast<PN_GLOBAL u32x4 *>(op) = *reinterpret_cast<u32x4 *>(ov);
                    } else {
                        reinterpret_cast<PN_GLOBAL u32x4 *>(op)[0] = reinterpret_cast<u32x4 *>(ov)[0];
                        reinterpret_cast<PN_GLOBAL u32x4 *>(op)[1] = reinterpret_cast<u32x4 *>(ov)[1];
                    }
                } else {
                    for (int k = 0; k < 8; ++k)
                        if (co + k < P.cout) op[k] = (T)v[k];
                }
            }
        }
        if (P.out_nchw) {
            const int ncol = min(BC, P.cout - cur.cb * BC);
            const size_t hw = (size_t)P.Ho * Wo;
            for (int i = tid; i < ncol * cur.npix; i += 256) {
                const int col = i / cur.npix, slot = i - col * cur.npix;
                const int co = cur.cb * BC + col;
                const int ry = (int)(((float)slot + 0.5f) * cur.inv_wc);
                const int rx = cur.ox0 + (slot - ry * cur.Wc);
                float v = *reinterpret_cast<const float *>(smem + slot * ROWB + col * 4);
                v = pn_activate(v, act, co, P.yolo_naf);
                ((PN_GLOBAL float *)P.out_nchw)[((size_t)cur.b * P.cout + co) * hw + (size_t)(cur.oy0 + ry) * Wo + rx] = v;
            }
        }
        if (!has_next) break;
        __syncthreads();                                 // the output tile has been read back: LDS is free again
        cur = nxt;
        make_baddr(cur);
    }
}

template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
static int conv_launch_persist(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    auto kern = conv_mfma_persist_kernel<PREC, KS, STRIDE, PITCH, CFG>;
    const size_t lds = L.lds_bytes + 16;             // + the tile hand-over word
    if (lds > 160 * 1024) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "conv halo tile needs %zu B of LDS", lds);
    if (lds > 48 * 1024) {
        static size_t configured = 0;
        if (configured < lds) {
            PN_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = lds;
        }
    }
    const int nblk = L.total_tiles < 2 * ctx->num_cus ? L.total_tiles : 2 * ctx->num_cus;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, stream, L.probs_dev, L.nprob, L.total_tiles, L.counter, (int)L.lds_bytes);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

#undef PN_CASE
#define PN_CASE(PREC, KS, ST, PITCH, CFG)                                                       \
    if (L.prec == PREC && L.ks == KS && L.stride == ST && L.pitch == PITCH && L.cfg == CFG)     \
        return L.persist ? conv_launch_persist<PREC, KS, ST, PITCH, CFG>(ctx, L, stream)        \
                         : conv_launch_one<PREC, KS, ST, PITCH, CFG>(ctx, L, stream);
