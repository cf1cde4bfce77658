// host/twostage.hpp -- part of the host side of libcsmp.so (included by csmp.hip, in order; ONE translation unit):
// OMP with replacement, the stepwise-regression object, srr, rmp, foba, br.
// ------------------------------------------------------------------------------------------ OMP with replacement
// The OMPR object (struct OMPR, src/twostage.jl:110-132) as the host keeps it: the support x (sorted, with its values), the residual
// norm, and the on-device QR + explicit inverse that belong to it.  csmp_ompr drives one through begin / acquire / update; the
// step-level functor (csmp_solver_begin(CSMP_ALGO_OMPR), host/steps_twostage.hpp) hands the same three calls to the host language.
struct OmprJob {
    csmp_ctx* ctx = nullptr;
    int64_t k = 0;
    std::vector<int64_t> xi;
    std::vector<double> xv;
    double resnorm = 0.0;
    bool use_downdate = false, tmode = false, screened = false;
    int unc_seen = 0;  // DevState::uncertain seen so far (the screened sweeps count up in it)
    std::vector<double> cs, call;
    // exchanges on the inverse Gram matrix (csmp_swap.hpp): on from the acquisition until an exchange fails its guard
    bool gram = false;
    int64_t gram_exchanges = 0;              // accepted since the solve began
    static constexpr int kSwapRefresh = 32;  // ... and every so many of them H, c, x are rebuilt from the factorisation
    std::vector<int> slot;  // the atom in every slot of H (device: s.sel)
    double *dG = nullptr, *dUpart = nullptr, *dU = nullptr, *dHp = nullptr, *dC = nullptr, *dInfo = nullptr, *dRpart = nullptr, *dN2 = nullptr;
    int* dMeta = nullptr;         // the exchange the chain performs: (1, leaving, joining, slot), csmp_swap.hpp
    unsigned* dCounter = nullptr; // k_swap_ufin's workgroup count (zero between launches)
    int nchk = 0, nchm = 0, nshare = 0;

    // H = (A_S'A_S)^-1 = T T' from the explicit inverse factor the acquisition built, c = A_S'b, x in slot order (s.bwd_coef)
    template <typename TA>
    int gram_begin_t() {
        Solver& s = ctx->s;
        const int kk = (int)k, M = (int)ctx->M;
        nchk = (kk + kSwapChunk - 1) / kSwapChunk;
        nchm = (kk + kResChunk - 1) / kResChunk;
        nshare = (M + 255) / 256;
        const size_t need = (size_t)(kk + 2) + (size_t)nchk * kk + 3 * (size_t)kk + 8 + (size_t)nchm * M + (size_t)nshare;
        if (!s.swapH) CHECK(dmalloc(ctx, &s.swapH, (size_t)s.kcap * s.kcap));
        if (s.swapv_cap < need) {
            dfree(s.swapv);
            CHECK(dmalloc(ctx, &s.swapv, need));
            s.swapv_cap = need;
        }
        dG = s.swapv;
        dUpart = dG + kk + 2;
        dU = dUpart + (size_t)nchk * kk;
        dHp = dU + kk;
        dC = dHp + kk;
        dInfo = dC + kk;
        dRpart = dInfo + 8;
        dN2 = dRpart + (size_t)nchm * M;
        dMeta = reinterpret_cast<int*>(dInfo + 4);          // (info uses three of its eight doubles)
        dCounter = reinterpret_cast<unsigned*>(dInfo + 6);
        HIPCHECK(hipMemsetAsync(dInfo, 0, 8 * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(k_swap_init, dim3((kk * kk + 255) / 256), dim3(256), 0, ctx->stream, (const double*)s.T, s.kcap, kk, s.swapH);
        hipLaunchKernelGGL((k_swap_dots<TA, double>), dim3((kk + 3) / 4), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M, (const int*)s.sel, kk,
                           (const double*)s.b, (const double*)s.b, 0, dC, (const int*)nullptr);
        HIPCHECK(hipGetLastError());
        slot.assign((size_t)kk, 0);
        HIPCHECK(hipMemcpyAsync(slot.data(), s.sel, (size_t)kk * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        gram = true;
        return CSMP_OK;
    }
    // the exchange dMeta names, queued on the stream (every kernel of it returns at once unless dMeta[0] == 1)
    template <typename TA>
    int gram_chain_t() {
        Solver& s = ctx->s;
        const int kk = (int)k, M = (int)ctx->M;
        const int ncommit = (kk * kk + 255) / 256, nrb = (M + 255) / 256;
        hipLaunchKernelGGL((k_swap_dots<TA, TA>), dim3((kk + 2 + 3) / 4), dim3(256), 0, ctx->stream, (const TA*)ctx->dA, ctx->ld, M, (const int*)s.sel, kk,
                           (const TA*)ctx->dA, (const double*)s.b, 2, dG, (const int*)dMeta);
        hipLaunchKernelGGL(k_swap_ufin, dim3(nchk), dim3(256), (size_t)kk * sizeof(int), ctx->stream, (const double*)s.swapH, s.kcap, kk, (const int*)dMeta,
                           (const double*)dG, dUpart, nchk, dCounter, dU, s.bwd_coef, dC, s.sel, dHp, dInfo, s.out_idx, s.out_val, s.out_nnz,
                           ctx->tune_swap_refuse ? 2.0 : 1e-6);
        hipLaunchKernelGGL(k_swap_commit_res<TA>, dim3(ncommit + nrb * nchm), dim3(256), 0, ctx->stream, s.swapH, s.kcap, kk, (const int*)dMeta, (const double*)dU,
                           (const double*)dHp, (const double*)dInfo, ncommit, (const TA*)ctx->dA, ctx->ld, M, (const int*)s.sel, (const double*)s.bwd_coef, nrb,
                           dRpart);
        hipLaunchKernelGGL(k_swap_rsum, dim3(nshare), dim3(256), 0, ctx->stream, (const double*)dRpart, nchm, M, (const double*)s.b, s.r, (const double*)dInfo, dN2,
                           (const int*)dMeta);
        HIPCHECK(hipGetLastError());
        return CSMP_OK;
    }
    int gram_chain() { return ctx->dtype == CSMP_F32 ? gram_chain_t<float>() : gram_chain_t<double>(); }
    // what the chain left: the pieces of one landing
    struct SwapLanding {
        std::vector<int64_t> hi;
        std::vector<double> hv, n2;
        double info[4] = {0, 0, 0, 0};
        int meta[4] = {0, 0, 0, 0};
    };
    int gram_land_add(PinFetch& f, SwapLanding& L) {
        Solver& s = ctx->s;
        const int kk = (int)k;
        L.hi.assign((size_t)kk, 0);
        L.hv.assign((size_t)kk, 0.0);
        L.n2.assign((size_t)nshare, 0.0);
        CHECK(f.add(L.hi.data(), s.out_idx, (size_t)kk * 8));
        CHECK(f.add(L.hv.data(), s.out_val, (size_t)kk * 8));
        CHECK(f.add(L.n2.data(), dN2, (size_t)nshare * 8));
        CHECK(f.add(L.info, dInfo, 32));
        CHECK(f.add(L.meta, dMeta, 16));
        return CSMP_OK;
    }
    size_t gram_land_bytes() const { return (size_t)k * 16 + (size_t)nshare * 8 + 128; }
    // the exchange went through: x, the slot's owner and ||r|| are the exchange's
    void gram_accept(const SwapLanding& L) {
        slot[(size_t)L.meta[3]] = L.meta[2];
        xi.assign(L.hi.begin(), L.hi.end());
        xv.assign(L.hv.begin(), L.hv.end());
        double t = 0.0;
        for (double v : L.n2) t += v;
        resnorm = std::sqrt(t);
        gram_exchanges += 1;
    }
    // H, c and x are UPDATED by every exchange (rank-one corrections), never re-derived: rounding accumulates over many exchanges on
    // an ill-conditioned support, and the per-exchange guard (sigma > 1e-6 a'a) sees a near-dependent atom, not drift.  The
    // reference solves against b afresh after every exchange (ldiv!!, src/twostage.jl:176).  So every kSwapRefresh accepted
    // exchanges the support is factorised from its columns again and H, c, x, r rebuilt from that (about one iteration's time).
    // The residual norm the stop rule compares is the one the exchanges carried: the refresh changes no decision by itself.
    int gram_refresh_if_due() {
        if (!gram || gram_exchanges == 0 || gram_exchanges % kSwapRefresh != 0) return CSMP_OK;
        const double keep = resnorm;
        CHECK(gram_leave());
        resnorm = keep;
        return ctx->dtype == CSMP_F32 ? gram_begin_t<float>() : gram_begin_t<double>();
    }
    // one exchange named by the host: `leaving` out, `joining` in; *refused: the guard did not hold and nothing was changed
    int gram_swap(int leaving, int joining, bool* refused) {
        const int kk = (int)k;
        const int p = (int)(std::find(slot.begin(), slot.end(), leaving) - slot.begin());
        if (p >= kk) return fail(ctx, CSMP_ESTATE, "ompr: the leaving atom is not in the support");
        void* pv = nullptr;
        CHECK(pin_get(ctx, 2, 16, &pv));
        int* m = (int*)pv;
        m[0] = 1; m[1] = leaving; m[2] = joining; m[3] = p;
        hipLaunchKernelGGL(k_put_ints, dim3(1), dim3(256), 0, ctx->stream, (const int*)m, 4, dMeta);
        HIPCHECK(hipGetLastError());
        CHECK(gram_chain());
        SwapLanding L;
        {
            PinFetch f(ctx);
            CHECK(f.begin(gram_land_bytes()));
            CHECK(gram_land_add(f, L));
            CHECK(f.wait());
        }
        *refused = L.info[2] != 0.0;
        if (!*refused) {
            gram_accept(L);
            CHECK(gram_refresh_if_due());
        }
        return CSMP_OK;
    }
    // the guard refused an exchange: back to the factorisation of the CURRENT support (as the acquisition builds it); the
    // rotation path takes over for the rest of the solve
    int gram_leave() {
        gram = false;
        std::vector<int> cols(xi.begin(), xi.end());
        CHECK(ls_on_columns(ctx, cols));
        CHECK(launch_tinv_build(ctx));
        return fetch_sorted_t(ctx, xi, xv);
    }

    // OMPR(A, b, k) (:124-132): buffers, b on the device, empty support, r = b
    int begin(csmp_ctx* c, const void* b, int b_dtype, int64_t kk) {
        ctx = c;
        k = kk;
        use_downdate = k <= kTMaxCols;  // (beyond: refactorise instead)
        tmode = use_downdate;           // explicit inverse next to R (csmp_tinv.hpp)
        if (use_downdate) CHECK(solver_fit_for_removal(ctx, (int)k));
        CHECK(solver_ensure(ctx, (int)k, (int)k));
        ctx->s.begun = false;
        if (use_downdate) CHECK(del_ensure(ctx));
        if (tmode) CHECK(tinv_ensure(ctx));
        CHECK(upload_b(ctx, b, b_dtype));
        // CSMP_OPT_SCREENED_SWEEP: the sweeps read the image; the oblivious acquisition's top-k set and every update!'s arg-max are
        // certified (host/screened.hpp), the correlations on the support are computed exactly beside them; a sweep whose
        // selection could not be certified is repeated exactly on the spot (the host reads the state after every sweep anyway)
        screened = screened_on(ctx) && k <= 4096;
        if (screened) CHECK(screened_ensure(ctx));
        xi.clear();
        xv.clear();
        cs.assign((size_t)k, 0.0);
        unc_seen = 0;
        resnorm = 0.0;
        gram = false;
        gram_exchanges = 0;
        return CSMP_OK;
    }
    // oblivious_acquisition!(P, x, k) on an empty x (src/matchingpursuit.jl:207-216, called at src/twostage.jl:190): the k atoms best
    // correlated with b, least squares on them; then norm(residual!(P, x)) (:192)
    int acquire() {
        Solver& s = ctx->s;
        std::vector<int> top((size_t)k);
        bool scr = screened;
        for (int attempt = 0; attempt < 2; ++attempt) {
            int flag = 0;
            if (scr) {
                CHECK(sp_select_screened(ctx, (int)k));
                HIPCHECK(hipMemcpyAsync(&flag, s.scr_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            } else {
                CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
                CHECK(launch_topS(ctx, (int)k));
            }
            HIPCHECK(hipMemcpyAsync(top.data(), s.cands, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            if (!scr) break;
            ctx->scr_solves += 1;
            if (!flag) break;
            ctx->scr_fallbacks += 1;
            scr = false;
        }
        std::sort(top.begin(), top.end());
        CHECK(ls_on_columns(ctx, top));
        if (tmode) {
            CHECK(launch_tinv_build(ctx));
            CHECK(fetch_sorted_t(ctx, xi, xv));
            if ((int64_t)xi.size() == k)  // (a full support: the exchanges run on the inverse Gram matrix from here on)
                CHECK(ctx->dtype == CSMP_F32 ? gram_begin_t<float>() : gram_begin_t<double>());
        } else {
            CHECK(fetch_sorted(ctx, xi, xv));
        }
        return residual_norm(ctx, &resnorm);
    }
    // the exchange on the factorisation (the rotation path, or a fresh factorisation beyond its capacity); x and, in explicit-
    // inverse mode, ||r|| come back in one synchronisation
    int exchange_on_factor(int leaving, int cand, bool* have_norm) {
        Solver& s = ctx->s;
        if (use_downdate) {
            // remove_column! + add_column! (:172-176) as a Givens down-date and a Gram-Schmidt append
            // (one launch names both atoms: the one that leaves -> its position, the one that joins -> the append's list)
            hipLaunchKernelGGL(k_swap_prep, dim3(1), dim3(256), 0, ctx->stream, (const int*)s.sel, (const DevState*)s.st, leaving, cand,
                               s.delpos, s.cands, s.ncands);
            HIPCHECK(hipGetLastError());
            if (tmode)
                CHECK(launch_delete_t(ctx));
            else
                CHECK(launch_delete(ctx));
            CHECK(launch_append(ctx, 2, 0, 0));
            if (tmode) CHECK(launch_tinv_append(ctx));
        } else {
            std::vector<int> cols;
            for (int64_t a : xi)
                if (a != leaving) cols.push_back((int)a);
            cols.insert(std::lower_bound(cols.begin(), cols.end(), cand), cand);
            CHECK(ls_on_columns(ctx, cols));  // :178
        }
        if (tmode) {
            CHECK(fetch_sorted_t(ctx, xi, xv, &resnorm));  // :178 and :196 in one synchronisation
            *have_norm = true;
        } else {
            CHECK(fetch_sorted(ctx, xi, xv));
        }
        return CSMP_OK;
    }
    // update!(P::OMPR, x) with eta = 1 (:134-180) and the norm(residual!(P, x)) the driver takes after it (:196)
    int update() {
        Solver& s = ctx->s;
        bool have_norm = false, changed = false;
        // Ar = x + A'r, arg-max over atoms outside the support
        std::vector<int> cur(xi.begin(), xi.end());
        DevState hs;
        // On the inverse Gram matrix with exact sweeps the "which entry leaves" decision is taken on the device too (k_ompr_pick) and
        // the exchange is queued behind it: ONE landing per update! carries the decision and its outcome.
        const bool fast = gram && !screened;
        SwapLanding L;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool scr = screened && attempt == 0;
            if (scr) {
                {  // the support goes up through a kernel that reads the page-locked staging buffer (no staged copy from pageable memory)
                    void* pcv = nullptr;
                    CHECK(pin_get(ctx, 2, cur.size() * 4 + 16, &pcv));
                    memcpy(pcv, cur.data(), cur.size() * 4);
                    hipLaunchKernelGGL(k_put_ints, dim3(((int)cur.size() + 255) / 256), dim3(256), 0, ctx->stream, (const int*)pcv, (int)cur.size(), s.cands);
                    HIPCHECK(hipGetLastError());
                }
                CHECK(ompr_sweep_screened(ctx, s.cands, (int)k));
            } else {
                // (the sorted support is on the device already: the index list the last k_emit_sorted wrote)
                CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
                hipLaunchKernelGGL(k_ompr_pick, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.pval, (const int*)s.pidx, ctx->sweep_grid,
                                   (const double*)s.cvec, s.st, (const int64_t*)s.out_idx, (int)k, s.coef, (const double*)s.out_val, (const int*)s.sel,
                                   fast ? dMeta : (int*)nullptr);
                HIPCHECK(hipGetLastError());
                if (fast) CHECK(gram_chain());
            }
            {
                PinFetch f(ctx);
                CHECK(f.begin((size_t)k * 8 + sizeof hs + 16 + (fast ? gram_land_bytes() : 0)));
                CHECK(f.add(cs.data(), s.coef, (size_t)k * 8));
                CHECK(f.add(&hs, s.st, sizeof hs));
                if (fast) CHECK(gram_land_add(f, L));
                CHECK(f.wait());
            }
            if (!scr) break;
            ctx->scr_solves += 1;
            const bool degenerate = std::binary_search(xi.begin(), xi.end(), (int64_t)hs.cand);
            if (hs.uncertain == unc_seen && !degenerate) break;  // certified, and the arg-max is a new atom
            if (hs.uncertain != unc_seen) ctx->scr_fallbacks += 1;
            unc_seen = hs.uncertain;  // (repeat with the exact sweep: it also leaves the correlation vector the degenerate case scans)
        }
        if (fast && L.meta[0] == 0) return CSMP_OK;  // no candidate, or the candidate itself was the smallest entry: nothing changed (:171)
        if (fast && L.meta[0] == 1) {
            if (L.info[2] == 0.0) {
                gram_accept(L);
                return gram_refresh_if_due();
            }
            // the guard refused: the same exchange on the factorisation of the current support, which carries the rest of the solve
            CHECK(gram_leave());
            CHECK(exchange_on_factor(L.meta[1], L.meta[2], &have_norm));
            if (!have_norm) CHECK(residual_norm(ctx, &resnorm));
            return CSMP_OK;
        }
        // (fast with L.meta[0] == 2: the arg-max lies inside the support -- nothing was queued; the host scans below)
        int64_t cand = hs.cand;
        double ccand = hs.cval;
        if (std::binary_search(xi.begin(), xi.end(), cand)) {
            // degenerate: the overall arg-max lies inside the support; scan the correlations on the host
            call.resize((size_t)ctx->N);
            HIPCHECK(hipMemcpy(call.data(), s.cvec, (size_t)ctx->N * 8, hipMemcpyDeviceToHost));
            cand = -1;
            double m = 0.0;
            for (int64_t j = 0; j < ctx->N; ++j) {
                if (std::binary_search(xi.begin(), xi.end(), j)) continue;
                const double f = std::fabs(call[j]);
                if (f > m) {  // strict '>' from m = 0: first maximum, none if everything is zero (:139-155)
                    m = f;
                    cand = j;
                }
            }
            if (cand >= 0) ccand = call[cand];
        } else if (!(std::fabs(ccand) > 0.0)) {
            cand = -1;
        }
        if (cand >= 0) {
            // x[i] = NaN; x.nzval = Ar[x.nzind]; drop the first entry of smallest magnitude (:158-169)
            const size_t pos = (size_t)(std::lower_bound(xi.begin(), xi.end(), cand) - xi.begin());
            size_t jmin = 0;
            double vmin = 0.0;
            for (size_t t = 0; t <= xi.size(); ++t) {
                const double v = t == pos ? ccand : (t < pos ? xv[t] + cs[t] : xv[t - 1] + cs[t - 1]);
                if (t == 0 || std::fabs(v) < vmin) {
                    vmin = std::fabs(v);
                    jmin = t;
                }
            }
            if (jmin != pos) {  // qr_i != j (:171): the support really changes
                changed = true;
                const int leaving = (int)(jmin < pos ? xi[jmin] : xi[jmin - 1]);
                if (gram) {
                    bool refused = false;
                    CHECK(gram_swap(leaving, (int)cand, &refused));
                    if (!refused) return CSMP_OK;  // (x, r and ||r|| are the exchange's)
                    CHECK(gram_leave());
                }
                CHECK(exchange_on_factor(leaving, (int)cand, &have_norm));
            }
        }
        // :196.  An update! that changed nothing leaves x, hence residual!(P, x) and its norm, exactly where they were (the
        // reference recomputes the same number bit for bit): the norm held is the norm -- re-measuring it by another kernel's
        // summation order could differ in the last place and keep `oldnorm <= resnorm` from ending the loop where it ends there
        if (!have_norm && changed) CHECK(residual_norm(ctx, &resnorm));
        return CSMP_OK;
    }
};

// (the context owns its OMPR object through a type-erased pointer: csmp_ctx is declared before this file)
static OmprJob& ompr_job(csmp_ctx* ctx) {
    if (!ctx->omprjob) ctx->omprjob = std::shared_ptr<void>(new OmprJob, [](void* p) { delete (OmprJob*)p; });
    return *(OmprJob*)ctx->omprjob.get();
}

// ompr(A,b,k,delta;maxiter): src/twostage.jl:110-202, x starting empty.  The support is filled by
// oblivious_acquisition! (src/matchingpursuit.jl:207-216); every update! (:134-180) is one sweep +
// arg-max on the device, the tiny "which entry leaves" decision on k+1 numbers on the host, and --
// when the support changes -- remove_column! as a Givens down-date of the on-device QR
// (csmp_downdate.hpp) followed by the usual append (k > 4095: a fresh panel factorisation instead).
extern "C" int csmp_ompr(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int64_t* idx,
                         double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!b || k < 1) return fail(ctx, CSMP_EINVAL, "ompr: b == NULL or k < 1");
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    if (k > ctx->N || k > ctx->M) return fail(ctx, CSMP_ERANGE, "ompr: k exceeds size(A)");
    if (maxiter < 0) maxiter = ctx->M;  // :185
    HIPCHECK(hipSetDevice(ctx->dev));
    OmprJob& j = ompr_job(ctx);
    CHECK(j.begin(ctx, b, b_dtype, k));
    ctx->scr_lone = true;
    struct LoneReset {
        csmp_ctx* c;
        ~LoneReset() { c->scr_lone = false; }
    } lone_reset{ctx};
    CHECK(j.acquire());
    int64_t it = 0;
    while (it < maxiter) {  // :193
        const double oldnorm = j.resnorm;
        CHECK(j.update());
        ++it;
        if (j.resnorm <= delta || oldnorm <= j.resnorm) break;  // :197
    }
    for (size_t t = 0; t < j.xi.size(); ++t) {
        if (idx) idx[t] = j.xi[t];
        if (val) val[t] = j.xv[t];
    }
    if (nnz) *nnz = (int64_t)j.xi.size();
    if (iters) *iters = it;
    return CSMP_OK;
}

// ------------------------------------------------------------------------------------------ stepwise regression object
// StepwiseRegression = ForwardRegression (src/forward.jl:14-32) kept on the device and driven from the
// host: forward_step! (src/forward.jl:56-73) = one dictionary sweep + one append, backward_step!
// (src/backward.jl:51-67) = scores from T = R^-1 + column removal.  The OLS rescaling rho2 follows the
// support through rank-one corrections folded into the NEXT sweep: -<a,q>^2 for the column a forward
// step appended, +<a,q_drop>^2 for the direction a backward step rotated out (csmp_forward.hpp,
// NQ = 2), so a forward/backward pair streams the dictionary once.  The host reads the 48-byte control
// block after every step (it must know which steps changed the support).
struct Stepwise {
    struct Pend { const double* q; double sgn; };
    csmp_ctx* ctx = nullptr;
    std::vector<Pend> pend;  // corrections rho2 still lacks; q == nullptr: the last Q column (device look-up)
    bool unmark = false;     // delmeta[2] names an atom that left the support and needs its rho2 re-seeded
    bool rho_ready = false;  // rho2 has been initialised (|a_j|^2 at least)
    int n = 0;               // atoms in the support
    double last_max_d2 = 0.0;  // maximum(P.δ²) of the last forward step
    int last_added = -1, last_removed = -1;  // atoms moved by the last successful forward / backward step
    DevState hs;

    int read_state(int* also_int = nullptr, const int* also_dev = nullptr, double* also_dbl = nullptr, const double* dbl_dev = nullptr) {
        Solver& s = ctx->s;
        PinFetch f(ctx);
        CHECK(f.begin(sizeof hs + 32));
        if (also_int) CHECK(f.add(also_int, also_dev, sizeof(int)));
        if (also_dbl) CHECK(f.add(also_dbl, dbl_dev, sizeof(double)));
        CHECK(f.add(&hs, s.st, sizeof hs));
        return f.wait();
    }
    int clear_flags() {
        static const int zero = 0;
        HIPCHECK(hipMemcpyAsync(&ctx->s.st->done, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        return CSMP_OK;
    }
    FrPass pass_of(int update_only) const {
        const Solver& s = ctx->s;
        FrPass ps;
        ps.nq = rho_ready ? (int)pend.size() : -1;
        ps.update_only = update_only;
        if (pend.size() >= 1) { ps.q1 = pend[0].q; ps.s1 = pend[0].sgn; }
        if (pend.size() >= 2) { ps.q2 = pend[1].q; ps.s2 = pend[1].sgn; }
        ps.unmark = unmark ? s.delmeta + 2 : nullptr;  // (the direction that was rotated out is always the last one)
        return ps;
    }
    // buffers for at most kcap atoms, b uploaded, empty support
    int begin(csmp_ctx* c, const void* b, int b_dtype, int kcap) {
        ctx = c;
        CHECK(solver_fit_for_removal(ctx, kcap));
        CHECK(solver_ensure(ctx, kcap, kcap));
        CHECK(fr_ensure(ctx));
        CHECK(tinv_ensure(ctx));
        Solver& s = ctx->s;
        s.begun = false;
        CHECK(upload_b(ctx, b, b_dtype));
        HIPCHECK(hipMemsetAsync(s.tmeta, 0, 2 * sizeof(int), ctx->stream));
        pend.clear();
        unmark = false;
        rho_ready = false;
        n = 0;
        return CSMP_OK;
    }
    // forward_step!(P, x, max_eps, min_delta); guarded == false: update!(P::FR, x) (src/forward.jl:88-95)
    int forward(double max_eps, double min_d2, bool guarded, bool* ok) {
        Solver& s = ctx->s;
        const int skipF = STOP_EPS | STOP_STAG | STOP_FULL;
        if (!guarded) {
            max_eps = -HUGE_VAL;
            min_d2 = -1.0;
        }
        CHECK(launch_fr_pass(ctx, pass_of(0), max_eps, skipF));
        CHECK(launch_append(ctx, 3, 0, skipF, false, min_d2, s.fr_grid));
        CHECK(launch_tinv_append(ctx));
        CHECK(read_state());
        if (hs.done & skipF) {
            // the step failed.  A residual-norm stop returns before rho2 is touched; the other guards act
            // after the sweep, which has then consumed the pending corrections.
            if (!(hs.done & STOP_EPS)) {
                pend.clear();
                unmark = false;
                rho_ready = true;
                last_max_d2 = hs.cval;
            }
            CHECK(clear_flags());
            *ok = false;
            return CSMP_OK;
        }
        last_max_d2 = hs.cval;
        last_added = hs.cand;
        rho_ready = true;
        pend.clear();
        unmark = false;
        pend.push_back({nullptr, -1.0});
        n = hs.nsel;
        *ok = true;
        return CSMP_OK;
    }
    // forward_step!(P, x, 0, 0) and, if it went through, backward_step!(P, x, Inf, Inf) right behind it -- srr's iteration with
    // l = 1 (src/twostage.jl:21-26) -- as ONE queue of launches and ONE landing: the backward step is gated on the device
    // (k_bwd_pick: no stop flag of the forward step, n0 + 1 atoms), so the host does not have to see the forward step's outcome
    // before it queues it.  *norm2: ||r||^2 after the removal.
    int forward_backward(bool* fok, bool* bok, double* norm2) {
        Solver& s = ctx->s;
        const int skipF = STOP_EPS | STOP_STAG | STOP_FULL;
        const int n0 = n;
        *fok = *bok = false;
        CHECK(launch_fr_pass(ctx, pass_of(0), 0.0, skipF));
        CHECK(launch_append(ctx, 3, 0, skipF, false, 0.0, s.fr_grid));
        CHECK(launch_tinv_append(ctx));
        // (a forward step that went through leaves {the last column} pending: no q_drop among it, nothing to flush)
        CHECK(launch_tinv_solve(ctx));
        hipLaunchKernelGGL(k_bwd_pick, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.bwd, (const int*)s.sel,
                           (const DevState*)s.st, (const double*)s.r, (int)ctx->M, (double)HUGE_VAL, (double)HUGE_VAL, s.delpos, s.bwd_info,
                           (const double*)nullptr, skipF, n0 + 1);
        HIPCHECK(hipGetLastError());
        CHECK(launch_delete_t(ctx));
        hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal);
        HIPCHECK(hipGetLastError());
        CHECK(read_state(&last_removed, s.delmeta + 2, norm2, s.scal));
        if (hs.done & skipF) {  // the forward step failed (as in forward()); nothing was removed
            if (!(hs.done & STOP_EPS)) {
                pend.clear();
                unmark = false;
                rho_ready = true;
                last_max_d2 = hs.cval;
            }
            CHECK(clear_flags());
            return CSMP_OK;
        }
        *fok = true;
        last_max_d2 = hs.cval;
        last_added = hs.cand;
        rho_ready = true;
        pend.clear();
        unmark = false;
        pend.push_back({nullptr, -1.0});
        n = n0 + 1;
        if (hs.nsel == n) return CSMP_OK;  // (no finite score: every atom stays)
        for (Pend& e : pend)
            if (!e.q) e.q = s.qsave;  // the appended column has been rotated; k_tdel_apply kept a copy
        pend.push_back({s.qdrop, 1.0});
        unmark = true;
        n = hs.nsel;
        *bok = true;
        return CSMP_OK;
    }
    // applies the pending corrections now (needed before a second removal reuses the q_drop buffer)
    int flush() {
        if (pend.empty() && !unmark) return CSMP_OK;
        CHECK(launch_fr_pass(ctx, pass_of(1), 0.0, 0));
        pend.clear();
        unmark = false;
        return CSMP_OK;
    }
    // backward_step!(P, x, max_eps, max_delta); lace: LACE's candidate rule (least |x_i|)
    // norm2: ||r||^2 after the step rides in the same landing (srr measures it once per iteration, right after its last removal)
    int backward(double max_eps, double max_d2, bool* ok, bool lace = false, double* norm2 = nullptr) {
        Solver& s = ctx->s;
        *ok = false;
        if (n <= 0) return CSMP_OK;
        bool has_drop = false;
        for (const Pend& e : pend) has_drop |= e.q == s.qdrop;
        if (has_drop) CHECK(flush());
        CHECK(launch_tinv_solve(ctx));
        hipLaunchKernelGGL(k_bwd_pick, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.bwd, (const int*)s.sel,
                           (const DevState*)s.st, (const double*)s.r, (int)ctx->M, max_eps, max_d2, s.delpos, s.bwd_info,
                           lace ? (const double*)s.bwd_coef : (const double*)nullptr);
        HIPCHECK(hipGetLastError());
        CHECK(launch_delete_t(ctx));
        if (norm2) {
            hipLaunchKernelGGL(k_norm2, dim3(1), dim3(256), 0, ctx->stream, (const double*)s.r, (int)ctx->M, s.scal);
            HIPCHECK(hipGetLastError());
        }
        CHECK(read_state(&last_removed, s.delmeta + 2, norm2, s.scal));
        if (hs.nsel == n) return CSMP_OK;  // the thresholds (or the lack of a finite score) kept every atom
        for (Pend& e : pend)
            if (!e.q) e.q = s.qsave;  // the appended column has been rotated; k_tdel_apply kept a copy
        pend.push_back({s.qdrop, 1.0});
        unmark = true;
        n = hs.nsel;
        *ok = true;
        return CSMP_OK;
    }
    int result(int64_t* idx, double* val, int64_t* nnz) {
        std::vector<int64_t> xi;
        std::vector<double> xv;
        CHECK(fetch_sorted_t(ctx, xi, xv));
        for (size_t t = 0; t < xi.size(); ++t) {
            if (idx) idx[t] = xi[t];
            if (val) val[t] = xv[t];
        }
        if (nnz) *nnz = (int64_t)xi.size();
        return CSMP_OK;
    }
};

template <typename TA, bool VEC>
static hipError_t fr_rebuild_lds_t(csmp_ctx* ctx, int d0, int nd) {
    Solver& s = ctx->s;
    auto kern = k_fr_rebuild_lds<TA, VEC>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fr_rebuild_lds_bytes());
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((ctx->N + 127) / 128)), dim3(256), fr_rebuild_lds_bytes(), ctx->stream, (const TA*)ctx->dA, ctx->ld,
                       (int)ctx->M, ctx->N, (const double*)s.Q, s.ldq, d0, nd, s.rho2);
    return hipGetLastError();
}
static int launch_fr_rebuild_lds(csmp_ctx* ctx, int d0, int nd) {
    // 16-byte loads of a column's rows need the column starts on 16-byte boundaries
    const size_t esz = ctx->dtype == CSMP_F32 ? 4 : 8;
    const bool vec = ((uintptr_t)ctx->dA % 16 == 0) && ((size_t)ctx->ld * esz) % 16 == 0;
    hipError_t e;
    if (ctx->dtype == CSMP_F32) e = vec ? fr_rebuild_lds_t<float, true>(ctx, d0, nd) : fr_rebuild_lds_t<float, false>(ctx, d0, nd);
    else e = vec ? fr_rebuild_lds_t<double, true>(ctx, d0, nd) : fr_rebuild_lds_t<double, false>(ctx, d0, nd);
    HIPCHECK(e);
    return CSMP_OK;
}

static int stepwise_args(csmp_ctx* ctx, const void* b, const char* who) {
    if (!ctx) return CSMP_EINVAL;
    if (!b) return fail(ctx, CSMP_EINVAL, (std::string(who) + ": b == NULL").c_str());
    if (!ctx->dA) return fail(ctx, CSMP_ESTATE, "no dictionary set (csmp_set_dictionary)");
    return CSMP_OK;
}

// srr(A,b,k,delta; maxiter=4k, initialization, l): src/twostage.jl:3-33, x starting empty.  initialization 3
// (random_acquisition!, src/matchingpursuit.jl:195-204) takes its k atoms from `init` (sorted, distinct): the draw is the
// caller's -- the reference takes it from the host language's RNG.
static int srr_impl(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int initialization,
                    const std::vector<int>* init, int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    CHECK(stepwise_args(ctx, b, "srr"));
    if (k < 1 || l < 1) return fail(ctx, CSMP_EINVAL, "srr: k < 1 or l < 1");
    if (initialization != 1 && initialization != 2 && !(initialization == 3 && init))
        return fail(ctx, CSMP_EINVAL, "srr: initialization must be 1 (oblivious) or 2 (forward regression); 3 (random) through csmp_srr_from");
    if (k > ctx->N || k + l > ctx->M) return fail(ctx, CSMP_ERANGE, "srr: k exceeds size(A)");
    if (k + l > kTMaxCols) return fail(ctx, CSMP_ERANGE, "srr: k + l exceeds 4095");
    if (maxiter < 0) maxiter = 4 * k;  // :5
    HIPCHECK(hipSetDevice(ctx->dev));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, (int)(k + l)));
    Solver& s = ctx->s;
    if (initialization == 1 || initialization == 3) {
        std::vector<int> top((size_t)k);
        if (initialization == 1) {
            // oblivious_acquisition!(P, x, k): src/matchingpursuit.jl:207-216
            CHECK(launch_sweep(ctx, s.r, 0.0, 0, 0));
            CHECK(launch_topS(ctx, (int)k));
            HIPCHECK(hipMemcpyAsync(top.data(), s.cands, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHECK(hipStreamSynchronize(ctx->stream));
            std::sort(top.begin(), top.end());
        } else {
            top = *init;  // random_acquisition!(P, x, k): :195-204 -- the caller's draw, sorted
        }
        CHECK(ls_on_columns(ctx, top));
        // rho2_j = |a_j|^2 - |Q'a_j|^2 for the k columns just factorised: the norms, then four columns per pass
        FrPass p0;
        p0.nq = -1;
        p0.update_only = 1;
        CHECK(launch_fr_pass(ctx, p0, 0.0, 0));
        {
            // Q'A on the Float64 matrix cores, 128 directions per pass (csmp_forward.hpp, k_fr_rebuild)
            const int grid = (int)((ctx->N + 127) / 128);  // 4 waves x 32 atoms
            for (int64_t t = 0; t < k; t += 128) {
                const int nd = (int)std::min<int64_t>(128, k - t);
                if (nd > 64 && !ctx->tune_rebuild_direct) {  // most of a 128-direction pass is real work: stage the directions in the LDS
                    CHECK(launch_fr_rebuild_lds(ctx, (int)t, nd));
                    continue;
                }
                if (ctx->dtype == CSMP_F32)
                    hipLaunchKernelGGL(k_fr_rebuild<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)ctx->dA, ctx->ld,
                                       (int)ctx->M, ctx->N, (const double*)s.Q, s.ldq, (int)t, nd, s.rho2);
                else
                    hipLaunchKernelGGL(k_fr_rebuild<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)ctx->dA, ctx->ld,
                                       (int)ctx->M, ctx->N, (const double*)s.Q, s.ldq, (int)t, nd, s.rho2);
                HIPCHECK(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(k_mark_inf, dim3(1), dim3(256), 0, ctx->stream, s.rho2, (const int*)s.sel, (const DevState*)s.st);
        HIPCHECK(hipGetLastError());
        CHECK(launch_tinv_build(ctx));
        CHECK(P.read_state());
        P.n = P.hs.nsel;
        P.rho_ready = true;
        if (P.hs.done) CHECK(P.clear_flags());
    } else {
        // k times update!(P::FR, x) (:12-15; src/forward.jl:88-95): enqueued back to back, no host round trips
        const int skip = STOP_FULL | STOP_STAG;
        for (int64_t t = 0; t < k; ++t) {
            CHECK(launch_fr_sweep(ctx, t == 0, -HUGE_VAL, skip));
            CHECK(launch_append(ctx, 3, 0, skip, false, -1.0, s.fr_grid));
        }
        CHECK(launch_tinv_build(ctx));
        CHECK(P.read_state());
        P.n = P.hs.nsel;
        P.rho_ready = true;
        if (P.n > 0) P.pend.push_back({nullptr, -1.0});  // the last appended column has not reached rho2 yet
        if (P.hs.done) CHECK(P.clear_flags());
    }
    double resnorm = 0.0;
    CHECK(residual_norm(ctx, &resnorm));  // :18
    int64_t it = 0;
    while (it < maxiter) {  // :19
        const double oldnorm = resnorm;
        std::vector<int> added, removed;
        double n2 = -1.0;  // ||r||^2 as the last removal left it (k_norm2 in that step's own landing), if there was one
        if (l == 1 && P.n == k) {  // the usual iteration: one atom in, one out, queued together
            bool fok, bok;
            CHECK(P.forward_backward(&fok, &bok, &n2));
            if (fok) added.push_back(P.last_added);
            if (bok) removed.push_back(P.last_removed);
            else n2 = -1.0;
        } else {
            for (int64_t f = 0; f < l; ++f) {  // :21-23  forward_step!(P, x, 0, 0) || break
                bool ok;
                CHECK(P.forward(0.0, 0.0, true, &ok));
                if (!ok) break;
                added.push_back(P.last_added);
            }
        }
        while (P.n > k) {  // :24-26  backward_step!(P, x, Inf, Inf)
            bool ok;
            n2 = -1.0;
            CHECK(P.backward((double)HUGE_VAL, (double)HUGE_VAL, &ok, false, P.n == k + 1 ? &n2 : nullptr));
            if (!ok) break;
            removed.push_back(P.last_removed);
        }
        std::sort(added.begin(), added.end());
        std::sort(removed.begin(), removed.end());
        // An iteration that removed exactly the atoms it added left x where it was: the residual norm is the
        // old one (:27-28 then stops).  Measuring it instead would compare two roundings of the same number.
        if (added == removed)
            resnorm = oldnorm;
        else if (n2 >= 0.0 && P.n == k)
            resnorm = std::sqrt(n2);  // :27
        else
            CHECK(residual_norm(ctx, &resnorm));  // :27
        ++it;
        if (resnorm <= delta || oldnorm <= resnorm) break;  // :28-30
    }
    if (iters) *iters = it;
    return P.result(idx, val, nnz);
}

// ------------------------------------------------------------------------------------------ relevance matching pursuit, FoBa
// src/stepwise.jl (x starting empty): loops over the two steps of the object above.  kmax bounds the
// support the forward stage may build (at most 4095; the reference's only bound is size(A,1)): a
// forward stage that needs more atoms than that ends with CSMP_ERANGE rather than a truncated answer.
static int stepwise_cap(csmp_ctx* ctx, int64_t kmax, int* kcap) {
    const int64_t lim = std::min<int64_t>(std::min<int64_t>(ctx->M, ctx->N), kTMaxCols);
    if (kmax <= 0) kmax = lim;
    *kcap = (int)std::min<int64_t>(kmax, lim);
    return CSMP_OK;
}
static int stepwise_full(csmp_ctx* ctx, const Stepwise& P, int kcap) {
    if (P.n >= kcap && kcap < std::min<int64_t>(ctx->M, ctx->N))
        return fail(ctx, CSMP_ERANGE, "stepwise regression: the forward stage filled the support capacity (kmax, at most 4095 atoms)");
    return CSMP_OK;
}
// !(xt ≈ x): isapprox with Julia's default rtol = sqrt(eps) on the sparse vectors
static bool x_changed(const std::vector<int64_t>& i0, const std::vector<double>& v0, const std::vector<int64_t>& i1,
                      const std::vector<double>& v1) {
    double d2 = 0.0, na = 0.0, nb = 0.0;
    size_t i = 0, j = 0;
    while (i < i0.size() || j < i1.size()) {
        double a = 0.0, b = 0.0;
        if (j >= i1.size() || (i < i0.size() && i0[i] < i1[j])) a = v0[i++];
        else if (i >= i0.size() || i1[j] < i0[i]) b = v1[j++];
        else { a = v0[i++]; b = v1[j++]; }
        d2 += (a - b) * (a - b);
        na += a * a;
        nb += b * b;
    }
    return !(std::sqrt(d2) <= 1.4901161193847656e-08 * std::sqrt(std::max(na, nb)));
}

// rmp(A, b, delta, maxiter): src/stepwise.jl:5-26

extern "C" int csmp_srr(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, int initialization,
                        int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (initialization == 3) return fail(ctx, CSMP_EINVAL, "srr: initialization 3 (random) needs the drawn atoms: csmp_srr_from");
    return srr_impl(ctx, b, b_dtype, k, delta, maxiter, initialization, nullptr, l, idx, val, nnz, iters);
}

// srr with initialization = 3: init[0..k) are the k distinct atoms random_acquisition! would have drawn (any order)
extern "C" int csmp_srr_from(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, double delta, int64_t maxiter, const int64_t* init,
                             int64_t l, int64_t* idx, double* val, int64_t* nnz, int64_t* iters) {
    if (!ctx) return CSMP_EINVAL;
    if (!init || k < 1) return fail(ctx, CSMP_EINVAL, "srr_from: init == NULL or k < 1");
    std::vector<int> top((size_t)k);
    for (int64_t t = 0; t < k; ++t) {
        if (init[t] < 0 || init[t] >= ctx->N) return fail(ctx, CSMP_ERANGE, "srr_from: atom index out of range");
        top[t] = (int)init[t];
    }
    std::sort(top.begin(), top.end());  // sort!(ind) (:197)
    if (std::adjacent_find(top.begin(), top.end()) != top.end()) return fail(ctx, CSMP_EINVAL, "srr_from: duplicate atom");
    return srr_impl(ctx, b, b_dtype, k, delta, maxiter, 3, &top, l, idx, val, nnz, iters);
}
extern "C" int csmp_rmp_delta(csmp_ctx* ctx, const void* b, int b_dtype, double delta, int64_t maxiter, int64_t kmax, int64_t* idx,
                              double* val, int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "rmp"));
    if (maxiter < 0) maxiter = 1;
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    const double d2 = delta * delta;
    std::vector<int64_t> xi0, xi;
    std::vector<double> xv0, xv;
    for (int64_t it = 0; it < maxiter; ++it) {  // :10
        for (int64_t f = 0; f < ctx->M; ++f) {  // :12-14
            bool ok;
            CHECK(P.forward(0.0, d2, true, &ok));
            if (!ok) break;
        }
        CHECK(stepwise_full(ctx, P, kcap));
        CHECK(fetch_sorted_t(ctx, xi, xv));
        if (!x_changed(xi0, xv0, xi, xv)) break;  // :15
        xi0 = xi;
        xv0 = xv;
        for (int t = P.n; t >= 1; --t) {  // :18-20
            bool ok;
            CHECK(P.backward((double)HUGE_VAL, d2, &ok));
            if (!ok) break;
        }
        CHECK(fetch_sorted_t(ctx, xi, xv));
        if (!x_changed(xi0, xv0, xi, xv)) break;  // :21
        xi0 = xi;
        xv0 = xv;
    }
    return P.result(idx, val, nnz);
}

// rmp(A, b, k): src/stepwise.jl:32-43
extern "C" int csmp_rmp_k(csmp_ctx* ctx, const void* b, int b_dtype, int64_t k, int64_t kmax, int64_t* idx, double* val,
                          int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "rmp"));
    if (k < 0) return fail(ctx, CSMP_EINVAL, "rmp: k < 0");
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    for (int64_t f = 0; f < ctx->M; ++f) {  // :36-38
        bool ok;
        CHECK(P.forward(0.0, 0.0, true, &ok));
        if (!ok) break;
    }
    CHECK(stepwise_full(ctx, P, kcap));
    for (int t = P.n; t >= k + 1; --t) {  // :39-41
        bool ok;
        CHECK(P.backward((double)HUGE_VAL, (double)HUGE_VAL, &ok));
        if (!ok) break;
    }
    return P.result(idx, val, nnz);
}

// foba(A, b, delta): src/stepwise.jl:47-56
extern "C" int csmp_foba(csmp_ctx* ctx, const void* b, int b_dtype, double delta, int64_t kmax, int64_t* idx, double* val,
                         int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "foba"));
    HIPCHECK(hipSetDevice(ctx->dev));
    int kcap;
    CHECK(stepwise_cap(ctx, kmax, &kcap));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, kcap));
    const double d2 = delta * delta;
    for (int64_t f = 0; f < ctx->M; ++f) {  // :50
        bool ok;
        CHECK(P.forward(0.0, d2, true, &ok));  // :51
        if (!ok) break;
        const double half = std::sqrt(P.last_max_d2) / 2.0;  // :52-53
        for (;;) {
            CHECK(P.backward((double)HUGE_VAL, half * half, &ok));
            if (!ok) break;
        }
    }
    CHECK(stepwise_full(ctx, P, kcap));
    return P.result(idx, val, nnz);
}

// br(A,b,max_eps,max_delta,k) (src/backward.jl:27-35; fbr :154-162 is the same algorithm on the normal
// equations) and, with lace != 0, lace(A,b,eps,delta,k) (:233-270): the least-squares solution on ALL
// N <= M columns, then backward steps until k atoms are left or a threshold stops them.
extern "C" int csmp_br(csmp_ctx* ctx, const void* b, int b_dtype, double max_eps, double max_delta, int64_t k, int lace,
                       int64_t* idx, double* val, int64_t* nnz) {
    CHECK(stepwise_args(ctx, b, "br"));
    if (k < 0) return fail(ctx, CSMP_EINVAL, "br: k < 0");
    if (max_eps != max_eps || max_delta != max_delta) return fail(ctx, CSMP_EINVAL, "br: threshold is NaN");
    if (ctx->N > ctx->M) return fail(ctx, CSMP_ERANGE, "br: A needs to be overdetermined (size(A,2) <= size(A,1))");  // :218
    if (ctx->N > kTMaxCols) return fail(ctx, CSMP_ERANGE, "br: more than 4095 columns");
    HIPCHECK(hipSetDevice(ctx->dev));
    Stepwise P;
    CHECK(P.begin(ctx, b, b_dtype, (int)ctx->N));
    std::vector<int> all((size_t)ctx->N);
    for (int64_t j = 0; j < ctx->N; ++j) all[(size_t)j] = (int)j;
    CHECK(ls_on_columns(ctx, all));  // UpdatableQR(A); x = AiQR \ b   (:11,:30)
    CHECK(launch_tinv_build(ctx));
    CHECK(P.read_state());
    P.n = P.hs.nsel;
    if (P.hs.done) CHECK(P.clear_flags());
    const double d2 = max_delta * max_delta;
    for (int t = P.n; t >= k + 1; --t) {  // :31-33
        bool ok;
        CHECK(P.backward(max_eps, d2, &ok, lace != 0));
        if (!ok) break;
    }
    return P.result(idx, val, nnz);
}
