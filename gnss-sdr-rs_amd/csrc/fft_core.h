// fft_core.h — in-LDS mixed-radix Stockham FFT for gfx950 (wave64), register-in / register-out.
//
// Replaces the rustfft 6.1.0 plans used at src/acquisition/do_acquisition.rs:132-143,182,188
// (forward = sum x[n] e^{-j2pi kn/N}, inverse = e^{+...}, neither normalised).
//
// Shape: one workgroup of T threads owns one length-N transform whose working set lives in
// ONE LDS buffer (N complex f32 = 64 KB at N = 8000, so two workgroups share a CU's 160 KB).
//   pass 0     : inputs arrive in registers (the caller fuses its global loads / carrier mix /
//                x conj(code spectrum) into them), butterfly, scatter to LDS
//   pass 1..   : gather from LDS (stride-1 across lanes -> conflict-free ds_read_b64),
//                twiddle (one LDS-resident base twiddle per butterfly + power tree in registers),
//                butterfly in registers, scatter back
//   last pass  : outputs stay in registers in natural order (the caller fuses |.|^2 accumulation
//                or the coalesced store)
// Radices are compile-time; composite radices (16, 20, 25, 33, ...) are built in registers by
// Cooley-Tukey or Good-Thomas (coprime factors: no inner twiddles) with constexpr trig constants.
//
// The header is host/device portable on purpose: tests/cpu emulate the T threads and the barriers
// phase by phase with g++ to validate index maps and butterflies without a GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline __attribute__((always_inline))
#endif

namespace gm {

struct alignas(8) cf {
    float x, y;
};

GM_HD cf cf_make(float x, float y) { cf r; r.x = x; r.y = y; return r; }
GM_HD cf cf_add(cf a, cf b) { return cf_make(a.x + b.x, a.y + b.y); }
GM_HD cf cf_sub(cf a, cf b) { return cf_make(a.x - b.x, a.y - b.y); }
// complex multiply with 2 mul + 2 fma (FFT-internal: not a "faithful" reference op)
GM_HD cf cf_mul(cf a, cf b) {
    return cf_make(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
// multiply by -j (forward) or +j (inverse)
template <bool INV> GM_HD cf cf_mulj(cf a) {
    if constexpr (!INV) return cf_make(a.y, -a.x);
    else return cf_make(-a.y, a.x);
}

// ------------------------------------------------------------------ constexpr trigonometry
namespace ct {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double tsin(double x) {  // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 14; ++i) { term *= -x2 / double((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double tcos(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; ++i) { term *= -x2 / double((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
struct cs { double c, s; };
// cos/sin(2*pi*a/b), exact quadrant reduction in integers
constexpr cs cossin2pi(long a, long b) {
    a %= b;
    if (a < 0) a += b;
    long q = (4 * a) / b, rem = 4 * a - q * b;  // angle = q*pi/2 + (pi/2)*rem/b
    double c = 1.0, s = 0.0;
    if (rem != 0) {
        if (2 * rem <= b) { double f = (kPi / 2) * double(rem) / double(b); c = tcos(f); s = tsin(f); }
        else { double f = (kPi / 2) * double(b - rem) / double(b); c = tsin(f); s = tcos(f); }
    }
    switch (q) {
        case 0: return {c, s};
        case 1: return {-s, c};
        case 2: return {-c, -s};
        default: return {s, -c};
    }
}
constexpr long gcd(long a, long b) { return b == 0 ? a : gcd(b, a % b); }
constexpr long modinv(long a, long m) {  // a^{-1} mod m, gcd == 1
    a %= m;
    for (long x = 1; x < m; ++x) if ((a * x) % m == 1) return x;
    return 1;
}
}  // namespace ct

// w_b^a = exp(-/+ j 2 pi a / b) as a compile-time constant
template <bool INV, long A, long B> GM_HD cf wconst() {
    constexpr ct::cs v = ct::cossin2pi(A, B);
    constexpr float c = float(v.c);
    constexpr float s = float(INV ? v.s : -v.s);
    return cf_make(c, s);
}
// u * w_B^A with the trivial cases folded
template <bool INV, long A, long B> GM_HD cf mul_wconst(cf u) {
    constexpr long a = ((A % B) + B) % B;
    if constexpr (a == 0) return u;
    else if constexpr (4 * a == B) return cf_mulj<INV>(u);                   // -j / +j
    else if constexpr (2 * a == B) return cf_make(-u.x, -u.y);               // -1
    else if constexpr (4 * a == 3 * B) return cf_mulj<!INV>(u);              // +j / -j
    else return cf_mul(u, wconst<INV, a, B>());
}

// ------------------------------------------------------------------ small butterflies (in place, natural order)
template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<1, INV> { static GM_HD void run(cf (&)[1]) {} };

template <bool INV> struct Dft<2, INV> {
    static GM_HD void run(cf (&u)[2]) {
        cf a = u[0], b = u[1];
        u[0] = cf_add(a, b); u[1] = cf_sub(a, b);
    }
};

template <bool INV> struct Dft<3, INV> {
    static GM_HD void run(cf (&u)[3]) {
        constexpr float s3 = 0.86602540378443864676f;
        cf t1 = cf_add(u[1], u[2]);
        cf m = cf_make(__builtin_fmaf(-0.5f, t1.x, u[0].x), __builtin_fmaf(-0.5f, t1.y, u[0].y));
        cf d = cf_sub(u[1], u[2]);
        cf js = cf_mulj<INV>(cf_make(s3 * d.x, s3 * d.y));
        u[0] = cf_add(u[0], t1); u[1] = cf_add(m, js); u[2] = cf_sub(m, js);
    }
};

template <bool INV> struct Dft<4, INV> {
    static GM_HD void run(cf (&u)[4]) {
        cf t0 = cf_add(u[0], u[2]), t1 = cf_sub(u[0], u[2]);
        cf t2 = cf_add(u[1], u[3]), t3 = cf_mulj<INV>(cf_sub(u[1], u[3]));
        u[0] = cf_add(t0, t2); u[1] = cf_add(t1, t3); u[2] = cf_sub(t0, t2); u[3] = cf_sub(t1, t3);
    }
};

template <bool INV> struct Dft<5, INV> {
    static GM_HD void run(cf (&u)[5]) {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        cf t1 = cf_add(u[1], u[4]), t2 = cf_add(u[2], u[3]);
        cf t3 = cf_sub(u[1], u[4]), t4 = cf_sub(u[2], u[3]);
        cf a1 = cf_make(__builtin_fmaf(c2, t2.x, __builtin_fmaf(c1, t1.x, u[0].x)),
                        __builtin_fmaf(c2, t2.y, __builtin_fmaf(c1, t1.y, u[0].y)));
        cf a2 = cf_make(__builtin_fmaf(c1, t2.x, __builtin_fmaf(c2, t1.x, u[0].x)),
                        __builtin_fmaf(c1, t2.y, __builtin_fmaf(c2, t1.y, u[0].y)));
        cf b1 = cf_make(__builtin_fmaf(s2, t4.x, s1 * t3.x), __builtin_fmaf(s2, t4.y, s1 * t3.y));
        cf b2 = cf_make(__builtin_fmaf(-s1, t4.x, s2 * t3.x), __builtin_fmaf(-s1, t4.y, s2 * t3.y));
        cf jb1 = cf_mulj<INV>(b1), jb2 = cf_mulj<INV>(b2);
        u[0] = cf_make(u[0].x + t1.x + t2.x, u[0].y + t1.y + t2.y);
        u[1] = cf_add(a1, jb1); u[4] = cf_sub(a1, jb1);
        u[2] = cf_add(a2, jb2); u[3] = cf_sub(a2, jb2);
    }
};

// odd prime R by the symmetric real-coefficient form: (R-1)^2 real FMAs
template <int R, bool INV> struct DftPrime {
    template <int Q, int J> static GM_HD void acc(const cf (&a)[(R - 1) / 2 + 1], const cf (&b)[(R - 1) / 2 + 1],
                                                 cf& ca, cf& sb) {
        if constexpr (J <= (R - 1) / 2) {
            constexpr ct::cs v = ct::cossin2pi(long(J) * Q, R);
            constexpr float c = float(v.c), s = float(v.s);
            ca.x = __builtin_fmaf(c, a[J].x, ca.x); ca.y = __builtin_fmaf(c, a[J].y, ca.y);
            sb.x = __builtin_fmaf(s, b[J].x, sb.x); sb.y = __builtin_fmaf(s, b[J].y, sb.y);
            acc<Q, J + 1>(a, b, ca, sb);
        }
    }
    template <int Q> static GM_HD void outq(const cf u0, const cf (&a)[(R - 1) / 2 + 1],
                                            const cf (&b)[(R - 1) / 2 + 1], cf (&y)[R]) {
        if constexpr (Q <= (R - 1) / 2) {
            cf ca = u0, sb = cf_make(0.f, 0.f);
            acc<Q, 1>(a, b, ca, sb);
            // forward: y[q] = ca - j*sb ; y[R-q] = ca + j*sb   (w = c - j s)
            cf jsb = cf_mulj<INV>(sb);
            y[Q] = cf_add(ca, jsb); y[R - Q] = cf_sub(ca, jsb);
            outq<Q + 1>(u0, a, b, y);
        }
    }
    static GM_HD void run(cf (&u)[R]) {
        constexpr int H = (R - 1) / 2;
        cf a[H + 1], b[H + 1], y[R];
        cf s0 = u[0];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            a[j] = cf_add(u[j], u[R - j]); b[j] = cf_sub(u[j], u[R - j]);
            s0 = cf_add(s0, a[j]);
        }
        a[0] = b[0] = cf_make(0.f, 0.f);
        y[0] = s0;
        outq<1>(u[0], a, b, y);
#pragma unroll
        for (int j = 0; j < R; ++j) u[j] = y[j];
    }
};
template <bool INV> struct Dft<7, INV> { static GM_HD void run(cf (&u)[7]) { DftPrime<7, INV>::run(u); } };
template <bool INV> struct Dft<11, INV> { static GM_HD void run(cf (&u)[11]) { DftPrime<11, INV>::run(u); } };
template <bool INV> struct Dft<13, INV> { static GM_HD void run(cf (&u)[13]) { DftPrime<13, INV>::run(u); } };
template <bool INV> struct Dft<31, INV> { static GM_HD void run(cf (&u)[31]) { DftPrime<31, INV>::run(u); } };

// Cooley-Tukey in registers: R = A*B, n = n2 + B*n1, k = k1 + A*k2
template <int A, int B, bool INV> struct DftCT {
    template <int N2, int K1> static GM_HD void tw_row(cf (&t)[A]) {
        if constexpr (K1 < A) {
            t[K1] = mul_wconst<INV, long(N2) * K1, long(A) * B>(t[K1]);
            tw_row<N2, K1 + 1>(t);
        }
    }
    template <int N2> static GM_HD void step1(const cf (&u)[A * B], cf (&v)[B][A]) {
        if constexpr (N2 < B) {
            cf t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) t[n1] = u[N2 + B * n1];
            Dft<A, INV>::run(t);
            tw_row<N2, 1>(t);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) v[N2][k1] = t[k1];
            step1<N2 + 1>(u, v);
        }
    }
    static GM_HD void run(cf (&u)[A * B]) {
        cf v[B][A];
        step1<0>(u, v);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            cf t[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) t[n2] = v[n2][k1];
            Dft<B, INV>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) u[k1 + A * k2] = t[k2];
        }
    }
};

// Good-Thomas (prime factor) in registers: gcd(A,B) = 1, no inner twiddles
//   n = (B*n1 + A*n2) mod N ; k = (k1*B*(B^-1 mod A) + k2*A*(A^-1 mod B)) mod N
template <int A, int B, bool INV> struct DftPFA {
    static GM_HD void run(cf (&u)[A * B]) {
        constexpr int N = A * B;
        constexpr int EA = int(B * ct::modinv(B, A)), EB = int(A * ct::modinv(A, B));
        cf v[B][A], y[N];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
            cf t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) t[n1] = u[(B * n1 + A * n2) % N];
            Dft<A, INV>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) v[n2][k1] = t[k1];
        }
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            cf t[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) t[n2] = v[n2][k1];
            Dft<B, INV>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) y[(k1 * EA + k2 * EB) % N] = t[k2];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) u[k] = y[k];
    }
};

template <bool INV> struct Dft<8, INV> { static GM_HD void run(cf (&u)[8]) { DftCT<2, 4, INV>::run(u); } };
template <bool INV> struct Dft<10, INV> { static GM_HD void run(cf (&u)[10]) { DftPFA<2, 5, INV>::run(u); } };
template <bool INV> struct Dft<15, INV> { static GM_HD void run(cf (&u)[15]) { DftPFA<3, 5, INV>::run(u); } };
template <bool INV> struct Dft<16, INV> { static GM_HD void run(cf (&u)[16]) { DftCT<4, 4, INV>::run(u); } };
template <bool INV> struct Dft<20, INV> { static GM_HD void run(cf (&u)[20]) { DftPFA<4, 5, INV>::run(u); } };
template <bool INV> struct Dft<24, INV> { static GM_HD void run(cf (&u)[24]) { DftPFA<3, 8, INV>::run(u); } };
template <bool INV> struct Dft<25, INV> { static GM_HD void run(cf (&u)[25]) { DftCT<5, 5, INV>::run(u); } };
template <bool INV> struct Dft<32, INV> { static GM_HD void run(cf (&u)[32]) { DftCT<4, 8, INV>::run(u); } };
template <bool INV> struct Dft<33, INV> { static GM_HD void run(cf (&u)[33]) { DftPFA<3, 11, INV>::run(u); } };

// ------------------------------------------------------------------ two-stage (streaming) butterflies
// A radix-R butterfly is split at its Cooley-Tukey / Good-Thomas seam so that the caller can put the
// in-place barrier BETWEEN the stages and so that inputs are consumed as they are loaded and outputs
// stored as they are produced: peak live state is the R intermediate values, not inputs + outputs.
//   stage1(in, v)  : in(r) yields input r (already twiddled); B sub-DFTs of size A -> v[R]
//   stage2(v, out) : A sub-DFTs of size B; out(k, value) receives output k
// KIND 0 = leaf (whole DFT in stage 1), 1 = Cooley-Tukey, 2 = Good-Thomas.
template <int R> struct Fac { static constexpr int A = R, B = 1, KIND = 0; };
template <> struct Fac<8> { static constexpr int A = 2, B = 4, KIND = 1; };
template <> struct Fac<10> { static constexpr int A = 2, B = 5, KIND = 2; };
template <> struct Fac<15> { static constexpr int A = 3, B = 5, KIND = 2; };
template <> struct Fac<16> { static constexpr int A = 4, B = 4, KIND = 1; };
template <> struct Fac<20> { static constexpr int A = 4, B = 5, KIND = 2; };
template <> struct Fac<24> { static constexpr int A = 3, B = 8, KIND = 2; };
template <> struct Fac<25> { static constexpr int A = 5, B = 5, KIND = 1; };
template <> struct Fac<32> { static constexpr int A = 4, B = 8, KIND = 1; };
template <> struct Fac<33> { static constexpr int A = 3, B = 11, KIND = 2; };

template <int R, bool INV> struct Bfly {
    static constexpr int A = Fac<R>::A, B = Fac<R>::B, KIND = Fac<R>::KIND;
    static constexpr int EA = KIND == 2 ? int(B * ct::modinv(B, A)) : 0;
    static constexpr int EB = KIND == 2 ? int(A * ct::modinv(A, B)) : 0;

    template <int N2, int K1> static GM_HD void inner_tw(cf (&t)[A]) {
        if constexpr (K1 < A) {
            t[K1] = mul_wconst<INV, long(N2) * K1, long(R)>(t[K1]);
            inner_tw<N2, K1 + 1>(t);
        }
    }
    template <int N2, class In> static GM_HD void s1(In& in, cf (&v)[R]) {
        if constexpr (N2 < B) {
            cf t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) {
                if constexpr (KIND == 2) t[n1] = in((B * n1 + A * N2) % R);
                else t[n1] = in(N2 + B * n1);
            }
            Dft<A, INV>::run(t);
            if constexpr (KIND == 1) inner_tw<N2, 1>(t);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) v[N2 * A + k1] = t[k1];
            s1<N2 + 1>(in, v);
        }
    }
    template <class In> static GM_HD void stage1(In&& in, cf (&v)[R]) { s1<0>(in, v); }

    template <class Out> static GM_HD void stage2(const cf (&v)[R], Out&& out) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int k = 0; k < R; ++k) out(k, v[k]);
        } else {
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) {
                cf t[B];
#pragma unroll
                for (int n2 = 0; n2 < B; ++n2) t[n2] = v[n2 * A + k1];
                Dft<B, INV>::run(t);
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) {
                    if constexpr (KIND == 2) out((k1 * EA + k2 * EB) % R, t[k2]);
                    else out(k1 + A * k2, t[k2]);
                }
            }
        }
    }
};

// w^r for r = 0..R-1 from the table value w^1: w^r = G[r / BS] * B[r % BS] (baby-step / giant-step).
// Only BS-1 + ceil(R/BS)-1 powers stay live (7 complex registers at R = 20), every power is at most
// ~4 rounded products away from w^1, and they are multiplied on the fly as inputs arrive.
template <int R> struct TwPow {
    static constexpr int BS = R <= 4 ? R : (R <= 9 ? 3 : (R <= 16 ? 4 : (R <= 25 ? 5 : 6)));
    static constexpr int NG = (R + BS - 1) / BS;
    cf Bp[BS], G[NG];
    GM_HD void init(cf w1) {
        Bp[1 % BS] = w1;
#pragma unroll
        for (int b = 2; b < BS; ++b) Bp[b] = cf_mul(Bp[b / 2], Bp[b - b / 2]);
        if constexpr (NG > 1) {
            G[1] = cf_mul(Bp[BS / 2], Bp[BS - BS / 2]);
#pragma unroll
            for (int g = 2; g < NG; ++g) G[g] = cf_mul(G[g / 2], G[g - g / 2]);
        }
    }
    GM_HD cf apply(cf u, int r) const {   // r is a compile-time constant after unrolling
        const int g = r / BS, b = r % BS;
        if (r == 0) return u;
        if (g == 0) return cf_mul(u, Bp[b]);
        if (b == 0) return cf_mul(u, G[g]);
        return cf_mul(u, cf_mul(G[g], Bp[b]));
    }
};

// ------------------------------------------------------------------ the plan
template <int N_, int T_, int... Rs> struct Plan {
    static constexpr int N = N_, T = T_, NP = int(sizeof...(Rs));
    static constexpr int R[NP] = {Rs...};
    static constexpr int P(int s) { int p = 1; for (int i = 0; i < s; ++i) p *= R[i]; return p; }
    static constexpr int NB(int s) { return N / R[s]; }
    static constexpr int IT(int s) { return (NB(s) + T - 1) / T; }
    static constexpr int TWOFF(int s) { int o = 0; for (int i = 1; i < s; ++i) o += P(i); return o; }
    static constexpr int TW_TOTAL = TWOFF(NP);
    // pad the pass0 -> pass1 image by one element per R0 when R0 is even (a stride-R0 scatter would
    // otherwise land 16 lanes on few banks); later scatters are runs of P >= R0 contiguous elements
    static constexpr int PAD_Q = (R[0] % 2 == 0) ? R[0] : 0;
    static constexpr int LDS_ELEMS = N + (PAD_Q ? N / PAD_Q : 0);
    static constexpr int R0 = R[0], IT0 = IT(0), RL = R[NP - 1], ITL = IT(NP - 1);
    // correlation kernel: keep conj(code spectrum) in registers across the integrations loop
    // (IT0*R0 complex VGPR pairs) only when it does not push the kernel past its VGPR budget
    static constexpr bool KEEP_CODE = (IT0 * R0 + ITL * RL) <= 24;
    // workgroups per CU the LDS footprint admits (160 KB per CU), and the waves per SIMD that needs
    static constexpr int LDS_BYTES = 8 * (LDS_ELEMS + TW_TOTAL);
    static constexpr int WG_PER_CU = (2 * LDS_BYTES <= 160 * 1024 && T <= 1024) ? 2 : 1;
    static constexpr int WAVES_PER_EU = (WG_PER_CU * T / 64 + 3) / 4;
    static constexpr bool check() { int p = 1; for (int i = 0; i < NP; ++i) p *= R[i]; return p == N && NP >= 2; }
    // pairwise coprime radices: the transform can run as a prime-factor (Good-Thomas) algorithm ACROSS the passes — no twiddle
    // between them at all — on inputs stored in the order Pfa<PL>::in_slot gives (see Pfa below)
    static constexpr bool coprime() {
        for (int i = 0; i < NP; ++i)
            for (int j = i + 1; j < NP; ++j)
                if (ct::gcd(R[i], R[j]) != 1) return false;
        return true;
    }
    static constexpr bool COPRIME = coprime();
    static constexpr bool HYBRID = false;                     // (HybridPlan below is the other kind of plan)
    static_assert(check(), "radices must multiply to N and there must be >= 2 passes");
};

// host-side: fill the base-twiddle table (TW_TOTAL entries): pass s >= 1, k in [0,P(s)):
//   tw[TWOFF(s) + k] = exp(-/+ j 2 pi k / (P(s) * R[s]))
template <class PL> inline void fill_twiddles(cf* tw, bool inverse, double (*cosfn)(double), double (*sinfn)(double)) {
    for (int s = 1; s < PL::NP; ++s) {
        const int p = PL::P(s), pr = p * PL::R[s];
        for (int k = 0; k < p; ++k) {
            double a = 2.0 * ct::kPi * double(k) / double(pr);
            tw[PL::TWOFF(s) + k] = cf_make(float(cosfn(a)), float(inverse ? sinfn(a) : -sinfn(a)));
        }
    }
}

// Which lanes run a MIDDLE pass: butterfly b of pass S is run by lane b + ROT (lanes below ROT sit the pass out).  A
// workgroup's waves w and w + 4 share a SIMD; a pass with fewer butterflies than lanes leaves its top waves idle, so with
// ROT = 0 everywhere the SIMD of waves 0 / 4 carries two butterfly streams in EVERY pass (N = 8000: 5, 7 and 8 of 8 waves
// busy -> 6 / 5 / 5 / 4 wave-passes per SIMD and transform).  Rotating pass 1 by one wave (waves 1..7 instead of 0..6) makes
// that 5 / 5 / 5 / 5.  Pass 0 and the last pass keep lane = butterfly: their index maps are shared with the callers' loads
// and stores.  Specialise per plan (fft_plans.h); default: no rotation.
template <class PL> struct PlanRot { static constexpr int rot(int) { return 0; } };

// ------------------------------------------------------------------ per-thread phases of one transform
// Every thread of the workgroup calls, in order ('|' = workgroup barrier):
//   pass0_stage1  | pass0_stage2 | mid_stage1<1> | mid_stage2<1> | ... | last_stage1  last_stage2
//   (the barrier before pass0_stage2 only orders it after the PREVIOUS transform's last_stage1 reads)
// Element (it, r) of pass S belongs to butterfly b = tid + it*T and is input index b + r*NB(S);
// outputs of the last pass are output index b + r*NB(last).  The split into phases is also what lets
// the CPU emulation (tests/cpu) interleave "threads".
// Prime-factor form across the passes (PL::COPRIME).  The Stockham data flow above with every twiddle = 1 computes the NP-
// dimensional DFT over the digits (r_0 .. r_{NP-1}) of the storage index e = ((r_0) R_1 + r_1) R_2 + r_2 ...; for pairwise
// coprime radices that IS the length-N DFT of the sequence whose element i sits at the slot with digits r_s = i mod R_s
// (CRT on the input side), with output (q_0 .. q_{NP-1}) being element o = sum_s (N / R_s) q_s mod N (Good's map on the
// output side): i * o = sum_s i (N/R_s) q_s, and W_N^{i (N/R_s) q_s} = W_{R_s}^{(i mod R_s) q_s}.  The writer of the input
// (stage F, the code-spectrum re-layout) applies in_slot for free; the reader of the output needs out_index only where an
// index is reported (the argmax).  No twiddle table, no power tree, no LDS reads for twiddles.
template <class PL> struct Pfa {
    static constexpr int NP = PL::NP;
    static GM_HD int in_slot(int i) {
        int e = 0;
#pragma unroll
        for (int s = 0; s < NP; ++s) e = e * PL::R[s] + i % PL::R[s];
        return e;
    }
    // storage slot -> element index (the inverse of in_slot): e has the digits r_s = i mod R_s, so i = sum_s A_s r_s mod N with
    // A_s = (N/R_s) * ((N/R_s)^-1 mod R_s)  (A_s = 1 mod R_s, = 0 mod every other radix)
    static constexpr int crt_coeff(int s) { return int((long(PL::N / PL::R[s]) * ct::modinv(PL::N / PL::R[s], PL::R[s])) % PL::N); }
    static GM_HD int slot_to_index(int e) {
        long i = 0;
#pragma unroll
        for (int s = NP - 1; s >= 0; --s) {
            i += long(crt_coeff(s)) * (e % PL::R[s]);
            e /= PL::R[s];
        }
        return int(i % PL::N);
    }
    // (butterfly b of the last pass, its output q) -> element index; b = sum_{s < NP-1} q_s P(s)
    static GM_HD int out_index(int b, int q) {
        int o = (PL::N / PL::R[NP - 1]) * q, k = b;
#pragma unroll
        for (int s = 0; s < NP - 1; ++s) {
            o += (PL::N / PL::R[s]) * (k % PL::R[s]);
            k /= PL::R[s];
            o = o >= PL::N ? o - PL::N : o;
        }
        return o;
    }
};

template <class PL, bool INV, bool PFA = false> struct Fft {
    static constexpr int NP = PL::NP;
    static_assert(!PFA || PL::COPRIME, "the prime-factor form needs pairwise coprime radices");

    static GM_HD int map01(int e) {
        if constexpr (PL::PAD_Q != 0) return e + e / PL::PAD_Q;
        else return e;
    }
    // LDS index of input element e of pass S (only the pass0 -> pass1 image is padded)
    template <int S> static GM_HD int rd(int e) { return S == 1 ? map01(e) : e; }

    // pass 0: in(it, r) yields the caller's input element (fused global load / mix / x conj(code))
    template <class In> static GM_HD void pass0_stage1(cf (&v)[PL::IT0][PL::R0], In&& in, int tid) {
#pragma unroll
        for (int it = 0; it < PL::IT0; ++it) {
            const int b = tid + it * PL::T;
            if (b < PL::NB(0)) Bfly<PL::R0, INV>::stage1([&](int r) { return in(it, r); }, v[it]);
        }
    }
    static GM_HD void pass0_stage2(const cf (&v)[PL::IT0][PL::R0], cf* lds, int tid) {
        constexpr int R = PL::R0;
#pragma unroll
        for (int it = 0; it < PL::IT0; ++it) {
            const int b = tid + it * PL::T;
            if (b < PL::NB(0)) {
                // map01(b*R + k) == b*(R + pad) + k : one base register, constant offsets
                cf* dst = lds + b * (R + (PL::PAD_Q ? 1 : 0));
                Bfly<R, INV>::stage2(v[it], [&](int k, cf val) { dst[k] = val; });
            }
        }
    }

    // lane -> butterfly of pass S: rotated for middle passes (PlanRot), the identity for pass 0 and the last pass
    template <int S> static GM_HD int bfly_of(int tid, int it) {
        constexpr int ROT = (S >= 1 && S <= NP - 2) ? PlanRot<PL>::rot(S) : 0;
        static_assert(ROT >= 0 && PL::IT(S) * PL::T - ROT >= PL::NB(S), "rotation leaves butterflies without a lane");
        return tid + it * PL::T - ROT;
    }
    template <int S> static GM_HD void gather_stage1(cf (&v)[PL::IT(S)][PL::R[S]], const cf* lds, const cf* tw, int tid) {
        constexpr int R = PL::R[S], NB = PL::NB(S), P = PL::P(S);
#pragma unroll
        for (int it = 0; it < PL::IT(S); ++it) {
            const int b = bfly_of<S>(tid, it);
            if (unsigned(b) < unsigned(NB)) {
                if constexpr (PFA) {
                    Bfly<R, INV>::stage1([&](int r) { return lds[rd<S>(b + r * NB)]; }, v[it]);
                } else {
                    TwPow<R> w;
                    w.init(tw[PL::TWOFF(S) + (b % P)]);
                    Bfly<R, INV>::stage1([&](int r) { return w.apply(lds[rd<S>(b + r * NB)], r); }, v[it]);
                }
            }
        }
    }
    template <int S> static GM_HD void mid_stage1(cf (&v)[PL::IT(S)][PL::R[S]], const cf* lds, const cf* tw, int tid) {
        gather_stage1<S>(v, lds, tw, tid);
    }
    template <int S> static GM_HD void mid_stage2(const cf (&v)[PL::IT(S)][PL::R[S]], cf* lds, int tid) {
        constexpr int R = PL::R[S], NB = PL::NB(S), P = PL::P(S);
#pragma unroll
        for (int it = 0; it < PL::IT(S); ++it) {
            const int b = bfly_of<S>(tid, it);
            if (unsigned(b) < unsigned(NB)) {
                const int k = b % P;
                cf* dst = lds + (b - k) * R + k;
                Bfly<R, INV>::stage2(v[it], [&](int q, cf val) { dst[q * P] = val; });
            }
        }
    }
    static GM_HD void last_stage1(cf (&v)[PL::ITL][PL::RL], const cf* lds, const cf* tw, int tid) {
        static_assert(PL::P(NP - 1) == PL::NB(NP - 1), "last pass has P == N/R");
        gather_stage1<NP - 1>(v, lds, tw, tid);
    }
    // out(it, r, value): output index (tid + it*T) + r*NB(last)
    template <class Out> static GM_HD void last_stage2(const cf (&v)[PL::ITL][PL::RL], Out&& out, int tid) {
#pragma unroll
        for (int it = 0; it < PL::ITL; ++it) {
            const int b = tid + it * PL::T;
            if (b < PL::NB(NP - 1)) Bfly<PL::RL, INV>::stage2(v[it], [&](int q, cf val) { out(it, q, val); });
        }
    }
};

// u * W_M^{K r} (exponent sign by INV) for r known after unrolling: a compile-time table of the R constants, trivial cases folded
template <bool INV, int K, int M, int R> struct ConstTw {
    struct Arr { float c[R], s[R]; };
    static constexpr Arr make() {
        Arr a{};
        for (int r = 0; r < R; ++r) {
            const ct::cs v = ct::cossin2pi(long(r) * K, M);
            a.c[r] = float(v.c);
            a.s[r] = float(INV ? v.s : -v.s);
        }
        return a;
    }
    static constexpr Arr tab = make();
    static GM_HD cf mul(cf u, int r) {
        const int e = int((long(r) * K) % M);
        if (e == 0) return u;
        if (4 * e == M) return cf_mulj<INV>(u);
        if (2 * e == M) return cf_make(-u.x, -u.y);
        if (4 * e == 3 * M) return cf_mulj<!INV>(u);
        return cf_mul(u, cf_make(tab.c[r], tab.s[r]));
    }
};

// ------------------------------------------------------------------ hybrid prime-factor / Cooley-Tukey plan
// N = A * B with gcd(A, B) = 1, A = A1 * A2, B = B1 * B2, gcd(A1, B1) = 1.  Across the two dimensions the transform is a
// prime-factor (Good-Thomas) one — no twiddles between A and B — and inside each dimension a two-step Cooley-Tukey one:
//   pass 0  radix A1*B1 : the FIRST factors of both dimensions as one 2-D butterfly (the radix-(A1 B1) Good-Thomas butterfly)
//   pass 1  radix A2    : second factor of A, inputs twiddled by W_A^{n2 k1},  k1 in [0, A1)
//   pass 2  radix B2    : second factor of B, inputs twiddled by W_B^{m2 j1},  j1 in [0, B1)
// The twiddles depend on k1 / j1 only — A1 and B1 values — so the butterflies of a pass are dealt to the waves by that digit:
// every wave holds ONE k1 (pass 1) or ONE j1 (pass 2) and its twiddles are compile-time constants of its code path
// (mul_wconst: literals, trivial cases folded; k1 = 0 / j1 = 0 cost nothing).  No twiddle table, no power tree, no LDS reads
// for twiddles: N = 8000 = 125 * 64 as [20, 25, 16] spends ~740 wave-instructions per transform on twiddles where the
// plain [25, 20, 16] plan spends ~1960 (generation + application).
// Digits.  Storage slot of the input e = (r0 R1 + r1) R2 + r2 (what lane b0 = r1 R2 + r2 of pass 0 loads as element r0),
// r0 = (B1 n1 + A1 m1) mod R0 (Good's map inside the first butterfly), r1 = n2, r2 = m2, where the element's index i has
// iA = i mod A = A2 n1 + n2 and iB = i mod B = B2 m1 + m2 (CRT on the input side).  Outputs: q0 <-> (k1 = q0 mod A1,
// j1 = q0 mod B1), q1 = k2, q2 = j2; oA = k1 + A1 k2, oB = j1 + B1 j2, element o = (B oA + A oB) mod N (Good's map on the
// output side).  LDS images between the passes are laid out for conflict-free access by the lane order of the pass that
// reads them (see pos1 / pos2).
template <int N_, int T_, int A1_, int A2_, int B1_, int B2_> struct HybridPlan {
    static constexpr int N = N_, T = T_, NP = 3;
    static constexpr int A1 = A1_, A2 = A2_, B1 = B1_, B2 = B2_, A = A1_ * A2_, B = B1_ * B2_;
    static constexpr int R[3] = {A1_ * B1_, A2_, B2_};
    static constexpr int P(int s) { int p = 1; for (int i = 0; i < s; ++i) p *= R[i]; return p; }
    static constexpr int NB(int s) { return N / R[s]; }
    static constexpr int IT(int s) { return (NB(s) + T - 1) / T; }
    static constexpr int TWOFF(int) { return 0; }
    static constexpr int TW_TOTAL = 0, PAD_Q = 0;
    static constexpr int R0 = R[0], IT0 = IT(0), RL = R[2], ITL = IT(2);
    // ONE image: cell (row, group k1, column u = j1*R2 + r2) at row*STR1 + k1*GS1 + u, STR1 = R0*R2 + 1.
    //   after pass 0: row = r1 (input digit of pass 1);  after pass 1: row = q1 (its output digit) — pass 1 is IN PLACE PER LANE:
    //   lane u of group k1 reads the A2 cells of its own column and writes its A2 outputs back into them, so no barrier of any
    //   kind separates its reads from its writes (program order of one lane's LDS operations is enough).
    //   last pass: lane beta = k1*A2 + q1 of group j1 reads row q1, group k1, columns j1*R2 + r2.
    // The odd row stride spreads the last pass's reads (consecutive q1) over the banks; seven lanes of a 32-lane read group
    // meet a second address on their bank where beta crosses from one k1 to the next.
    static constexpr int GS1 = B1_ * R[2], STR1 = R[0] * R[2] + 1;
    static constexpr int LDS_ELEMS = A2_ * STR1;
    static constexpr int GW1 = GS1 / 64;                      // waves per k1 group of pass 1
    static constexpr bool P1_NEEDS_BARRIER = false;           // in place per lane
    static constexpr int GW2 = (A + 63) / 64;                 // waves per j1 group of the last pass
    static constexpr bool KEEP_CODE = false, COPRIME = false, HYBRID = true;
    static constexpr int LDS_BYTES = 8 * LDS_ELEMS;
    static constexpr int WG_PER_CU = (2 * LDS_BYTES <= 160 * 1024 && T <= 1024) ? 2 : 1;
    static constexpr int WAVES_PER_EU = (WG_PER_CU * T / 64 + 3) / 4;
    static_assert(A1_ * A2_ * B1_ * B2_ == N_ && ct::gcd(A1_ * A2_, B1_ * B2_) == 1 && ct::gcd(A1_, B1_) == 1, "N = A*B, A and B coprime");
    static_assert(IT(0) == 1 && IT(1) == 1 && IT(2) == 1, "one butterfly per lane and pass");
    static_assert(GS1 % 64 == 0 && A1_ * GW1 * 64 <= T_ && B1_ * GW2 * 64 <= T_, "wave groups must fit the workgroup");

    // element index -> storage slot, and back; (lane of the last pass, output q2) -> element index
    static GM_HD int in_slot(int i) {
        const int ia = i % A, ib = i % B;
        const int n1 = ia / A2_, n2 = ia % A2_, m1 = ib / B2_, m2 = ib % B2_;
        const int r0 = (B1_ * n1 + A1_ * m1) % R[0];
        return (r0 * R[1] + n2) * R[2] + m2;
    }
    static GM_HD int slot_to_index(int e) {
        const int m2 = e % R[2], n2 = (e / R[2]) % R[1], r0 = e / (R[1] * R[2]);
        // r0 = (B1 n1 + A1 m1) mod R0  ->  n1 = r0 * B1^-1 mod A1, m1 = r0 * A1^-1 mod B1
        const int n1 = (r0 * int(ct::modinv(B1_, A1_))) % A1_, m1 = (r0 * int(ct::modinv(A1_, B1_))) % B1_;
        const int ia = A2_ * n1 + n2, ib = B2_ * m1 + m2;
        // i = ia mod A, ib mod B
        constexpr long ca = long(B) * ct::modinv(B, A), cb = long(A) * ct::modinv(A, B);
        return int((ca * ia + cb * ib) % N);
    }
    // last pass: wave group j1, lane-in-group beta = k1*A2 + q1
    static GM_HD int last_j1(int tid) { return (tid >> 6) / GW2; }
    static GM_HD int last_beta(int tid) { return ((tid >> 6) % GW2) * 64 + (tid & 63); }
    static GM_HD bool last_active(int tid) { return last_j1(tid) < B1_ && last_beta(tid) < A; }
    static GM_HD int out_index(int tid, int q2) {
        const int beta = last_beta(tid), j1 = last_j1(tid);
        const int oa = beta / A2_ + A1_ * (beta % A2_), ob = j1 + B1_ * q2;
        return (B * oa + A * ob) % N;
    }
};

template <int N_, int T_, int A1, int A2, int B1, int B2, bool INV>
struct Fft<HybridPlan<N_, T_, A1, A2, B1, B2>, INV, false> {
    using PL = HybridPlan<N_, T_, A1, A2, B1, B2>;
    static constexpr int R0 = PL::R[0], R1 = PL::R[1], R2 = PL::R[2];

    template <class In> static GM_HD void pass0_stage1(cf (&v)[1][R0], In&& in, int tid) {
        if (tid < PL::NB(0)) Bfly<R0, INV>::stage1([&](int r) { return in(0, r); }, v[0]);
    }
    static GM_HD void pass0_stage2(const cf (&v)[1][R0], cf* lds, int tid) {
        if (tid < PL::NB(0)) {
            cf* dst = lds + (tid / R2) * PL::STR1 + tid % R2;           // lane b0 = r1 R2 + r2
            Bfly<R0, INV>::stage2(v[0], [&](int q0, cf val) { dst[(q0 % A1) * PL::GS1 + (q0 % B1) * R2] = val; });
        }
    }
    // pass 1: wave group k1 = wave / GW1, lane-in-group lambda = j1 R2 + r2
    template <int K1> static GM_HD void p1_s1(cf (&v)[R1], const cf* lds, int lam) {
        const cf* src = lds + K1 * PL::GS1 + lam;
        Bfly<R1, INV>::stage1([&](int r) { return ConstTw<INV, K1, PL::A, R1>::mul(src[r * PL::STR1], r); }, v);
    }
    template <int K1> static GM_HD void p1_disp(cf (&v)[R1], const cf* lds, int k1, int lam) {
        if (k1 == K1) p1_s1<K1>(v, lds, lam);
        else if constexpr (K1 + 1 < A1) p1_disp<K1 + 1>(v, lds, k1, lam);
    }
    template <int S> static GM_HD void mid_stage1(cf (&v)[1][R1], const cf* lds, const cf*, int tid) {
        static_assert(S == 1, "three passes");
        const int wave = tid >> 6, k1 = wave / PL::GW1, lam = (wave % PL::GW1) * 64 + (tid & 63);
        if (k1 < A1) p1_disp<0>(v[0], lds, k1, lam);
    }
    template <int S> static GM_HD void mid_stage2(const cf (&v)[1][R1], cf* lds, int tid) {
        const int wave = tid >> 6, k1 = wave / PL::GW1, lam = (wave % PL::GW1) * 64 + (tid & 63);
        if (k1 < A1) {
            cf* dst = lds + k1 * PL::GS1 + lam;                      // the lane's own column: output q1 replaces input r1 = q1
            Bfly<R1, INV>::stage2(v[0], [&](int q1, cf val) { dst[q1 * PL::STR1] = val; });
        }
    }
    // last pass: lane beta = k1*A2 + q1 of group J1 reads input r2 from row q1, group k1, column J1*R2 + r2
    template <int J1> static GM_HD void p2_s1(cf (&v)[R2], const cf* lds, int beta) {
        const int k1 = beta / A2, q1 = beta - k1 * A2;
        const cf* src = lds + q1 * PL::STR1 + k1 * PL::GS1 + J1 * R2;
        Bfly<R2, INV>::stage1([&](int r) { return ConstTw<INV, J1, PL::B, R2>::mul(src[r], r); }, v);
    }
    template <int J1> static GM_HD void p2_disp(cf (&v)[R2], const cf* lds, int j1, int beta) {
        if (j1 == J1) p2_s1<J1>(v, lds, beta);
        else if constexpr (J1 + 1 < B1) p2_disp<J1 + 1>(v, lds, j1, beta);
    }
    static GM_HD void last_stage1(cf (&v)[1][R2], const cf* lds, const cf*, int tid) {
        if (PL::last_active(tid)) p2_disp<0>(v[0], lds, PL::last_j1(tid), PL::last_beta(tid));
    }
    // out(0, q2, value): element PL::out_index(tid, q2)
    template <class Out> static GM_HD void last_stage2(const cf (&v)[1][R2], Out&& out, int tid) {
        if (PL::last_active(tid)) Bfly<R2, INV>::stage2(v[0], [&](int q, cf val) { out(0, q, val); });
    }
};

}  // namespace gm
