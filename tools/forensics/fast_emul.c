/*
 * fast_emul.c -- CPU emulation of the FAST arithmetic's CHOICES, one at a time, against the pinned oracle.
 *
 * ANALYSIS TOOLING (round 4, review item 1b): not product, not a test.  The default GPU arithmetic differs from the
 * reference evaluation in (S) the tridiagonal algorithm -- in-lane cyclic reduction + PCR on normalised rows + Cramer
 * pairs instead of pcreduce's order --, (A) an algebraically rewritten assembly with fused multiply-adds, (F) the field
 * update with Lambda folded into the diffusivities, (Q) the PL quadrature of per-node excesses.  This file takes the
 * oracle's restatement of the reference (oracle/trpl_oracle.c, included for its driver pieces) and swaps each of those in
 * SEPARATELY, with exact IEEE divisions everywhere, so that the distance of each to the oracle can be measured on the
 * CPU over the bench's window: which choice makes the state drift by ~1e-10 on stiff grids, where two evaluations of the
 * reference's own order (with / without FMA contraction) stay within 1e-12?
 *
 *   gcc -O2 -fPIC -std=c11 -ffp-contract=off -mfma -fopenmp -shared -o tools/forensics/libfastemul.so tools/forensics/fast_emul.c -lm
 * Driven by tools/forensics/run_emul.py.
 */
#include "../../oracle/trpl_oracle.c"

/* ---- solver variants: (ld, d, ud, B) -> x, all of length L; destroys its inputs ---- */

/* PCR in pcreduce's order of levels but on NORMALISED rows, multiply-adds fused (pcr.hpp: pcr64_levels on every row) */
static void solve_pcr_norm(double *ld, double *d, double *ud, double *B, double *x, double *buf, int N)
{
    double *nA = buf, *nC = buf + N, *nB = buf + 2 * N;
    int rf = 1;
    for (; N > 2 * rf; rf *= 2) {
        for (int i = 0; i < N; i++) { double r = 1.0 / d[i]; nA[i] = ld[i] * r; nC[i] = ud[i] * r; nB[i] = B[i] * r; }
        for (int i = 0; i < N; i++) {
            int m = (i - rf + N) % N, p = (i + rf) % N;            /* wrapped values meet exact zeros */
            double A = ld[i], C = ud[i];
            d[i] = fma(-C, nA[p], fma(-A, nC[m], d[i]));
            B[i] = fma(-C, nB[p], fma(-A, nB[m], B[i]));
            ld[i] = -(A * nA[m]);
            ud[i] = -(C * nC[p]);
        }
        /* (in-place update above reads nA.. snapshots only: fine) */
    }
    for (int i = 0; i < N; i++) {                                   /* Cramer, own unknown only (pcr.hpp: cr_pcr_solve) */
        int low = i < rf, o = low ? i + rf : i - rf;
        double c_own = low ? ud[i] : ld[i], c_oth = low ? ld[o] : ud[o];
        x[i] = fma(-c_own, B[o], B[i] * d[o]) / fma(-c_own, c_oth, d[i] * d[o]);
    }
}

/* NRL in-lane cyclic-reduction levels (stride 1, 2, .. NR/2), PCR on the N/NR rows that are left, back-substitution */
static void solve_cr_pcr(double *ld, double *d, double *ud, double *B, double *x, double *buf, int N, int NR, int cramer)
{
    for (int H = 1; H < NR; H *= 2) {
        for (int q = H; q < N; q += 2 * H) { double r = 1.0 / d[q]; ld[q] *= r; ud[q] *= r; B[q] *= r; }
        for (int p = 0; p < N; p += 2 * H) {
            double a = ld[p], c = ud[p];
            double aL = p >= H ? ld[p - H] : 0.0, cL = p >= H ? ud[p - H] : 0.0, bL = p >= H ? B[p - H] : 0.0;
            double aR = p + H < N ? ld[p + H] : 0.0, cR = p + H < N ? ud[p + H] : 0.0, bR = p + H < N ? B[p + H] : 0.0;
            d[p] = fma(-c, aR, fma(-a, cL, d[p]));
            B[p] = fma(-c, bR, fma(-a, bL, B[p]));
            ld[p] = -(a * aL);
            ud[p] = -(c * cR);
        }
    }
    const int M = N / NR;
    double *A = buf, *D = buf + M, *C = buf + 2 * M, *Bv = buf + 3 * M, *nA = buf + 4 * M, *nC = buf + 5 * M, *nB = buf + 6 * M, *X = buf + 7 * M;
    for (int m = 0; m < M; m++) { A[m] = ld[m * NR]; D[m] = d[m * NR]; C[m] = ud[m * NR]; Bv[m] = B[m * NR]; }
    int S = 1;
    for (; S < M / 2; S *= 2) {
        for (int m = 0; m < M; m++) { double r = 1.0 / D[m]; nA[m] = A[m] * r; nC[m] = C[m] * r; nB[m] = Bv[m] * r; }
        for (int m = 0; m < M; m++) {
            int dn = (m - S + M) % M, up = (m + S) % M;
            double a = A[m], c = C[m];
            D[m] = fma(-c, nA[up], fma(-a, nC[dn], D[m]));
            Bv[m] = fma(-c, nB[up], fma(-a, nB[dn], Bv[m]));
            A[m] = -(a * nA[dn]);
            C[m] = -(c * nC[up]);
        }
    }
    if (cramer) {
        for (int m = 0; m < M; m++) {
            int low = m < M / 2, o = low ? m + M / 2 : m - M / 2;
            double c_own = low ? C[m] : A[m], c_oth = low ? A[o] : C[o];
            X[m] = fma(-c_own, Bv[o], Bv[m] * D[o]) / fma(-c_own, c_oth, D[m] * D[o]);
        }
    } else {                                                       /* the reference's 2x2 elimination (:75-79) */
        for (int m = 0; m < M / 2; m++) {
            int o = m + M / 2;
            double k = C[m] / D[o];
            X[m] = (Bv[m] - Bv[o] * k) / (D[m] - A[o] * k);
            X[o] = (Bv[o] - A[o] * X[m]) / D[o];
        }
    }
    for (int m = 0; m < M; m++) x[m * NR] = X[m];
    for (int H = NR / 2; H >= 1; H /= 2)
        for (int q = H; q < N; q += 2 * H) {
            double xr = q + H < N ? x[q + H] : 0.0;
            x[q] = fma(-ud[q], xr, fma(-ld[q], x[q - H], B[q]));
        }
}

typedef struct { int solver, assembly, field, quad; } emul_cfg;
/* solver: 0 pcreduce (reference)  1 normalised PCR  2 CR x1 + PCR64  3 CR x2 + PCR32 (paired kernel)  4 as 3, reference 2x2
 * assembly: 0 reference  1 FAST rewrite;  field: 0 reference  1 Lambda folded;  quad: 0 reference sum  1 per-node excess */

static void emul_solve(const emul_cfg *cf, sysws *w, double *A2, double *A1, double *A0, double *bb, double *x, int L, double *xbuf)
{
    switch (cf->solver) {
    case 0: oracle_pcreduce(A2, A1, A0, bb, x, w->buffer, L); break;
    case 1: solve_pcr_norm(A2, A1, A0, bb, x, w->buffer, L); break;
    case 2: solve_cr_pcr(A2, A1, A0, bb, xbuf, w->buffer, L, 2, 1); memcpy(x, xbuf, sizeof(double) * L); break;
    case 3: solve_cr_pcr(A2, A1, A0, bb, xbuf, w->buffer, L, 4, 1); memcpy(x, xbuf, sizeof(double) * L); break;
    default: solve_cr_pcr(A2, A1, A0, bb, xbuf, w->buffer, L, 4, 0); memcpy(x, xbuf, sizeof(double) * L); break;
    }
}

/* FAST assembly of one equation (stepper_impl.hpp: assemble<LAY = 2>), exact reciprocals */
static void assemble_fast(int is_n, const double *mp, double a0, const double *Nk, const double *Pk, const double *Ek /*[L], E_L = 0*/,
                          const double *bU, double *lo, double *dg, double *up, double *bb, int L)
{
    const double N0 = mp[0], P0 = mp[1], DN = mp[2], DP = mp[3], rate = mp[4], sr0 = mp[5], srL = mp[6], CN = mp[7], CP = mp[8],
                 tauN = mp[9], tauP = mp[10];
    const double n0p0 = N0 * P0;
    const double D = is_n ? DN : DP, hD = is_n ? 0.5 * D : -0.5 * D;
    const double Cx = is_n ? CP : CN, tauO = is_n ? tauN : tauP;
    const double Co2 = 2.0 * (is_n ? CN : CP), Con0 = (is_n ? CN : CP) * n0p0, tVn0 = (is_n ? tauP : tauN) * n0p0;
    for (int i = 0; i < L; i++) {
        const double E = Ek[i], Ep = i + 1 < L ? Ek[i + 1] : 0.0;
        const double U = is_n ? Nk[i] : Pk[i], V = is_n ? Pk[i] : Nk[i];
        const double u_i = i == L - 1 ? 0.0 : fma(-hD, Ep, -D);
        const double l_i = i == 0 ? 0.0 : fma(hD, E, -D);
        const double tp = fma(Pk[i], tauN, Nk[i] * tauP);
        const double np_ = fma(Nk[i], Pk[i], -n0p0);
        const double inv = 1.0 / tp;
        const double V2 = V * V;
        const double X = fma(tauO, V2, tVn0) * (inv * inv);
        const double Y = fma(Cx, V2, fma(Co2, np_, Con0));
        const double s = fma(rate, V, X) + Y;
        const double t = fma(CP, Pk[i], fma(CN, Nk[i], rate)) + inv;
        up[i] = u_i; lo[i] = l_i;
        if (i > 0 && i < L - 1) dg[i] = fma(hD, E - Ep, a0 + 2.0 * D) + s;
        else {
            const double u_m = i == 0 ? 0.0 : fma(-hD, E, -D), l_p = i == L - 1 ? 0.0 : fma(hD, Ep, -D);
            dg[i] = a0 - u_m - l_p + s;
        }
        bb[i] = fma(-t, np_, fma(s, U, -bU[i]));
    }
    for (int e = 0; e < 2; e++) {                                  /* surfaces */
        const int i = e ? L - 1 : 0;
        const double sr = e ? srL : sr0, Ns = Nk[i], Ps = Pk[i];
        const double Vs = is_n ? Ps : Ns, Us = is_n ? Ns : Ps;
        const double inv = 1.0 / (Ns + Ps), g = sr * inv;
        const double dss = -(g * inv) * fma(Vs, Vs, n0p0);
        const double fs = fma(g, fma(Ns, Ps, -n0p0), dss * Us);
        dg[i] = fma(-1.0, dss, dg[i]);
        bb[i] = fma(-1.0, fs, bb[i]);
    }
}

/* reference assembly, copied call structure of it_sweep (oracle) but split per equation */
static void assemble_ref(int is_n, const double *mp, double a0, const double *Nk, const double *Pk, const double *Ek,
                         const double *bU, double *A2, double *A1, double *A0, double *bb, int L)
{
    const double N0 = mp[0], P0 = mp[1], DN = mp[2], DP = mp[3], rate = mp[4], sr0 = mp[5], srL = mp[6], CN = mp[7], CP = mp[8],
                 tauN = mp[9], tauP = mp[10];
    const double n0p0 = N0 * P0;
    A0[L - 1] = 0; A2[0] = 0;
    if (is_n) {
        for (int n = 1; n < L; n++) { A0[n - 1] = DN * (-Ek[n] / 2 - 1); A2[n] = DN * (+Ek[n] / 2 - 1); }
        for (int n = 0; n < L; n++) {
            double tp = Nk[n] * tauP + Pk[n] * tauN, np_ = Nk[n] * Pk[n] - n0p0;
            double ds = -rate * Pk[n] - (Pk[n] * tp - tauP * np_) / (tp * tp) - (CN * Nk[n] * Pk[n] + CP * (Pk[n] * Pk[n]) + CN * np_);
            A1[n] = a0 - A0[(n + L - 1) % L] - A2[(n + 1) % L] - ds;
            bb[n] = -(CN * Nk[n] + CP * Pk[n] + rate + 1 / tp) * np_ - ds * Nk[n] - bU[n];
        }
        double s0 = Nk[0] + Pk[0], sL = Nk[L - 1] + Pk[L - 1];
        double ds0 = -sr0 * (Pk[0] * Pk[0] + n0p0) / (s0 * s0), dsL = -srL * (Pk[L - 1] * Pk[L - 1] + n0p0) / (sL * sL);
        A1[0] -= ds0; A1[L - 1] -= dsL;
        bb[0] -= sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Nk[0];
        bb[L - 1] -= srL * (Nk[L - 1] * Pk[L - 1] - n0p0) / sL + dsL * Nk[L - 1];
    } else {
        for (int n = 1; n < L; n++) { A0[n - 1] = DP * (+Ek[n] / 2 - 1); A2[n] = DP * (-Ek[n] / 2 - 1); }
        for (int n = 0; n < L; n++) {
            double np_ = Nk[n] * Pk[n] - n0p0, tp = Nk[n] * tauP + Pk[n] * tauN;
            double ds = -rate * Nk[n] - (Nk[n] * tp - tauN * np_) / (tp * tp) - (CP * Nk[n] * Pk[n] + CN * (Nk[n] * Nk[n]) + CP * np_);
            A1[n] = a0 - A0[(n + L - 1) % L] - A2[(n + 1) % L] - ds;
            bb[n] = -(CN * Nk[n] + CP * Pk[n] + rate + 1 / tp) * np_ - ds * Pk[n] - bU[n];
        }
        double s0 = Nk[0] + Pk[0], sL = Nk[L - 1] + Pk[L - 1];
        double ds0 = -sr0 * (Nk[0] * Nk[0] + n0p0) / (s0 * s0), dsL = -srL * (Nk[L - 1] * Nk[L - 1] + n0p0) / (sL * sL);
        A1[0] -= ds0; A1[L - 1] -= dsL;
        bb[0] -= sr0 * (Nk[0] * Pk[0] - n0p0) / s0 + ds0 * Pk[0];
        bb[L - 1] -= srL * (Nk[L - 1] * Pk[L - 1] - n0p0) / sL + dsL * Pk[L - 1];
    }
}


/* assembly with the FAST rewrite switched on piece by piece (mask):  1 off-diagonals by one fma (hD E - D) instead of
 * D (E/2 - 1);  2 transport part of the diagonal from the closed formula hD (E_i - E_{i+1}) + 2 D instead of minus the two
 * neighbouring off-diagonal entries AS ROUNDED;  4 recombination terms rewritten (s, t, bb by fma);  8 surface terms */
static void assemble_mask(int mask, int is_n, const double *mp, double a0, const double *Nk, const double *Pk, const double *Ek,
                          const double *bU, double *A2, double *A1, double *A0, double *bb, int L)
{
    const double N0 = mp[0], P0 = mp[1], DN = mp[2], DP = mp[3], rate = mp[4], sr0 = mp[5], srL = mp[6], CN = mp[7], CP = mp[8],
                 tauN = mp[9], tauP = mp[10];
    const double n0p0 = N0 * P0;
    const double D = is_n ? DN : DP, sg = is_n ? 1.0 : -1.0, hD = sg * 0.5 * D;
    const double Co = is_n ? CN : CP, Cx = is_n ? CP : CN, tauV = is_n ? tauP : tauN, tauO = is_n ? tauN : tauP;
    A0[L - 1] = 0; A2[0] = 0;
    for (int n = 1; n < L; n++) {
        if (mask & 1) { A0[n - 1] = fma(-hD, Ek[n], -D); A2[n] = fma(hD, Ek[n], -D); }
        else          { A0[n - 1] = D * (-sg * Ek[n] / 2 - 1); A2[n] = D * (+sg * Ek[n] / 2 - 1); }
    }
    for (int n = 0; n < L; n++) {
        const double U = is_n ? Nk[n] : Pk[n], V = is_n ? Pk[n] : Nk[n];
        double tr;                                               /* a0 - A0[n-1] - A2[n+1] */
        if ((mask & 2) && n > 0 && n < L - 1) tr = fma(hD, Ek[n] - Ek[n + 1], a0 + 2.0 * D);
        else tr = a0 - A0[(n + L - 1) % L] - A2[(n + 1) % L];
        if (mask & 4) {
            const double tp = fma(Pk[n], tauN, Nk[n] * tauP), np_ = fma(Nk[n], Pk[n], -n0p0), inv = 1.0 / tp, V2 = V * V;
            const double X = fma(tauO, V2, tauV * n0p0) * (inv * inv);
            const double Y = fma(Cx, V2, fma(2.0 * Co, np_, Co * n0p0));
            const double sN = fma(rate, V, X) + Y;
            const double t = fma(CP, Pk[n], fma(CN, Nk[n], rate)) + inv;
            A1[n] = tr + sN;
            bb[n] = fma(-t, np_, fma(sN, U, -bU[n]));
        } else {
            const double tp = Nk[n] * tauP + Pk[n] * tauN, np_ = Nk[n] * Pk[n] - n0p0;
            const double ds = -rate * V - (V * tp - tauV * np_) / (tp * tp) - (Co * Nk[n] * Pk[n] + Cx * (V * V) + Co * np_);
            A1[n] = tr - ds;
            bb[n] = -(CN * Nk[n] + CP * Pk[n] + rate + 1 / tp) * np_ - ds * U - bU[n];
        }
    }
    for (int e = 0; e < 2; e++) {
        const int i = e ? L - 1 : 0;
        const double sr = e ? srL : sr0, Ns = Nk[i], Ps = Pk[i], Vs = is_n ? Ps : Ns, Us = is_n ? Ns : Ps;
        if (mask & 8) {
            const double inv = 1.0 / (Ns + Ps), g = sr * inv;
            const double dss = -(g * inv) * fma(Vs, Vs, n0p0);
            const double fs = fma(g, fma(Ns, Ps, -n0p0), dss * Us);
            A1[i] = fma(-1.0, dss, A1[i]); bb[i] = fma(-1.0, fs, bb[i]);
        } else {
            const double s0 = Ns + Ps, ds0 = -sr * (Vs * Vs + n0p0) / (s0 * s0);
            A1[i] -= ds0;
            bb[i] -= sr * (Ns * Ps - n0p0) / s0 + ds0 * Us;
        }
    }
}

static void emul_sweep(const emul_cfg *cf, sysws *w, const double *mp, double a0, double *errN, double *errP, double *xbuf)
{
    const int L = w->L;
    const double DN = mp[2], DP = mp[3], Lambda = mp[11];
    double *Nk = w->Nk, *Pk = w->Pk, *Ek = w->Ek, *bb = w->bb, *A0 = w->A0, *A1 = w->A1, *A2 = w->A2;
    for (int eq = 0; eq < 2; eq++) {
        const int is_n = eq == 0;
        double *U = is_n ? Nk : Pk;
        if (cf->assembly >= 16) assemble_mask(cf->assembly - 16, is_n, mp, a0, Nk, Pk, Ek, is_n ? w->bN : w->bP, A2, A1, A0, bb, L);
        else if (cf->assembly) assemble_fast(is_n, mp, a0, Nk, Pk, Ek, is_n ? w->bN : w->bP, A2, A1, A0, bb, L);
        else              assemble_ref(is_n, mp, a0, Nk, Pk, Ek, is_n ? w->bN : w->bP, A2, A1, A0, bb, L);
        *(is_n ? errN : errP) = oracle_norm2(A0, A1, A2, bb, U, w->buffer, L);
        emul_solve(cf, w, A2, A1, A0, bb, U, L, xbuf);
    }
    if (cf->field) {
        const double LDP = Lambda * DP, LDN = Lambda * DN, hLDP = 0.5 * LDP, hLDN = 0.5 * LDN;
        for (int n = 1; n < L; n++) {
            double A = fma(hLDP, Pk[n] + Pk[n - 1], fma(hLDN, Nk[n] + Nk[n - 1], a0));
            double b = fma(LDP, Pk[n] - Pk[n - 1], fma(-LDN, Nk[n] - Nk[n - 1], -w->bE[n]));
            Ek[n] = b / A;
        }
    } else {
        for (int n = 1; n < L; n++) {
            A1[n] = Lambda * (DP * (Pk[n] + Pk[n - 1]) + DN * (Nk[n] + Nk[n - 1])) / 2 + a0;
            bb[n] = Lambda * (DP * (Pk[n] - Pk[n - 1]) - DN * (Nk[n] - Nk[n - 1])) - w->bE[n];
            Ek[n] = bb[n] / A1[n];
        }
    }
}

/* pvSim + tEvol with the choices of cfg swapped in; plI [S][T+1] fp64, plT = 1; optional state dump of the last step */
/* history storage (round-3 review, "Next" 4).  0: the reference's ring of fp64 levels.
 * 1: fp32 DIFFERENCES FROM THE NEWEST LEVEL, D_j = U^{k+1-j} - U^k (j = 2..5); every row of the BDF table sums to zero, so
 *    bU = -a0 U^k + sum_j a_j D_j; per step D'_2 = -delta, D'_j = D_{j-1} - delta with delta = U^{k+1} - U^k in fp64, rounded
 *    to fp32 on storing.
 * 2: fp32 BACKWARD differences W_1 = U^k - U^{k-1}, W_2 = W_1 - W_1', ... (W_1 .. W_4), bU from the binomial expansion of
 *    the levels; per step W'_1 = delta, W'_m = W'_{m-1} - W_{m-1}.   3: as 2 with W_1 kept in fp64. */
static int g_hist = 0;
void emul_set_history(int mode) { g_hist = mode; }

static void hist_rhs(int mode, const double *a, const double *Uk, const float *Df, const double *W1d, int n, int ld, double *bU)
{
    /* Df[m * ld + n], m = 0..3 */
    if (mode == 1) {
        bU[n] = -a[0] * Uk[n] + a[2] * (double)Df[0 * ld + n] + a[3] * (double)Df[1 * ld + n] + a[4] * (double)Df[2 * ld + n] + a[5] * (double)Df[3 * ld + n];
    } else {
        /* U^{k-1} = U - W1, U^{k-2} = U - 2 W1 + W2, U^{k-3} = U - 3 W1 + 3 W2 - W3, U^{k-4} = U - 4 W1 + 6 W2 - 4 W3 + W4 */
        const double w1 = mode == 3 ? W1d[n] : (double)Df[0 * ld + n], w2 = Df[1 * ld + n], w3 = Df[2 * ld + n], w4 = Df[3 * ld + n];
        const double c1 = -(a[2] + 2 * a[3] + 3 * a[4] + 4 * a[5]), c2 = a[3] + 3 * a[4] + 6 * a[5], c3 = -(a[4] + 4 * a[5]), c4 = a[5];
        bU[n] = -a[0] * Uk[n] + c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4;
    }
}
static void hist_push(int mode, const double *Unew, const double *Uold, float *Df, double *W1d, int n, int ld)
{
    const double delta = Unew[n] - Uold[n];
    if (mode == 1) {
        const float d1 = Df[0 * ld + n], d2 = Df[1 * ld + n], d3 = Df[2 * ld + n];
        Df[3 * ld + n] = (float)((double)d3 - delta);
        Df[2 * ld + n] = (float)((double)d2 - delta);
        Df[1 * ld + n] = (float)((double)d1 - delta);
        Df[0 * ld + n] = (float)(-delta);
    } else {
        const double w1 = mode == 3 ? W1d[n] : (double)Df[0 * ld + n];
        const float w2 = Df[1 * ld + n], w3 = Df[2 * ld + n];
        const float n2 = (float)(delta - w1);
        const float n3 = (float)((double)n2 - (double)w2);
        const float n4 = (float)((double)n3 - (double)w3);
        W1d[n] = delta; Df[0 * ld + n] = (float)delta; Df[1 * ld + n] = n2; Df[2 * ld + n] = n3; Df[3 * ld + n] = n4;
    }
}

int emul_pvsim(const double *matpar, long S, double length, double time_, int L, long T, int tol, int MAX,
               const double *inipar, double *plI, int64_t *iters_total, int solver, int assembly, int field, int quad,
               int nthreads)
{
    const emul_cfg cf = { solver, assembly, field, quad };
    double scales[12], dx3, plnorm;
    oracle_scales(length, time_, L, T, scales, &dx3, &plnorm);
    const double TOL = pow(10.0, -(double)tol);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long y = 0; y < S; y++) {
        sysws w;
        if (ws_alloc(&w, L)) continue;
        double *xbuf = (double *)malloc(sizeof(double) * L);
        const int hist = g_hist, LE = L + 1;
        float *Df = (float *)calloc((size_t)12 * LE, sizeof(float));     /* N, P, E: 4 stored differences each */
        double *W1 = (double *)calloc((size_t)3 * LE, sizeof(double));
        double mp[12];
        for (int i = 0; i < 12; i++) mp[i] = matpar[y * 12 + i] * scales[i];
        for (int n = 0; n < L; n++) { double dN = inipar[n] * dx3; w.N[n] = mp[0] + dN; w.P[n] = mp[1] + dN; }
        int64_t itot = 0;
        for (long t = 0; t <= T; t++) {
            double a[6];
            bdf_coeffs(t, a);
            int kp = pymod6(t + 1), k = pymod6(t);
            int ko[4] = { pymod6(t - 1), pymod6(t - 2), pymod6(t - 3), pymod6(t - 4) };
            it_begin(&w, a, k, ko);
            if (hist) {
                for (int n = 0; n < L; n++) {
                    hist_rhs(hist, a, w.N + k * L, Df, W1, n, LE, w.bN);
                    hist_rhs(hist, a, w.P + k * L, Df + 4 * LE, W1 + LE, n, LE, w.bP);
                    hist_rhs(hist, a, w.E + k * LE, Df + 8 * LE, W1 + 2 * LE, n, LE, w.bE);
                }
            }
            int iters;
            for (iters = 0; iters < MAX; iters++) {
                double eN, eP;
                emul_sweep(&cf, &w, mp, a[0], &eN, &eP, xbuf);
                if (eN < TOL && eP < TOL) break;
            }
            it_end(&w, kp);
            if (hist) {
                for (int n = 0; n < L; n++) {
                    hist_push(hist, w.N + kp * L, w.N + k * L, Df, W1, n, LE);
                    hist_push(hist, w.P + kp * L, w.P + k * L, Df + 4 * LE, W1 + LE, n, LE);
                    hist_push(hist, w.E + kp * LE, w.E + k * LE, Df + 8 * LE, W1 + 2 * LE, n, LE);
                }
            }
            itot += (iters < MAX ? iters : MAX - 1) + 1;
            const double *Nc = w.N + k * L, *Pc = w.P + k * L;
            double Sum;
            if (cf.quad) { Sum = 0; for (int n = 0; n < L; n++) Sum += fma(Nc[n], Pc[n], -(mp[0] * mp[1])); }
            else { Sum = -(double)L * (mp[0] * mp[1]); for (int n = 0; n < L; n++) Sum += Nc[n] * Pc[n]; }
            plI[y * (T + 1) + t] = mp[4] * Sum / plnorm;
        }
        if (iters_total) iters_total[y] = itot;
        free(xbuf); free(Df); free(W1);
        ws_free(&w);
    }
    return 0;
}
