// rtx_device.hpp -- per-pixel arithmetic of the ray-trace hot path for gfx950 device code.
//
// Every function follows the reference's fp32 operation order exactly (SURVEY.md Appendix A;
// citations are file:line under ConsoleProject/) so that results are bit-identical to an IEEE
// evaluation of the reference: no FMA contraction (-ffp-contract=off), correctly rounded
// sqrt and division (hipcc's default), no reassociation, denormals kept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx {

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// MyMath.h:60-63, 74-77, 88-92
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 mulf(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
// MyMath.cu:5-8: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// ---- correctly rounded 1.0f/x and sqrtf(x) in fewer instructions than hipcc's generic expansions.
//
// The path needs the IEEE results bit for bit.  hipcc expands 1.0f/x into v_div_scale x2, v_rcp,
// six FMAs/fmas and v_div_fixup, and sqrtf into v_sqrt plus a two-candidate residual test: about 10
// and 16 instructions, most of them guarding ranges this path never visits.  Inside a wide safe
// range the sequences below take 3 and 5; outside it they fall back to the generic expansion.
// Equality with the generic results is verified EXHAUSTIVELY, for all 2^32 inputs, on the GPU
// (tests/gpu_checks/math_check.hip, tests/test_gpu_math.py): that test is the proof.
constexpr float kSafeLo = 0x1.0p-60f, kSafeHi = 0x1.0p+60f;

// generic IEEE expansions, kept out of line: they are the rare path
__device__ __attribute__((noinline)) float rcp_generic(float x) { return 1.0f / x; }
__device__ __attribute__((noinline)) float sqrt_generic(float x) { return sqrtf(x); }

// x in [2^-60, 2^60] (either sign): hardware estimate (1 ulp) and one Newton step.  Exhaustively
// equal to 1.0f/x on the whole range (bare v_rcp differs on 215 million inputs; with the step: none).
__device__ __forceinline__ float rcp_fast(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// x in [2^-60, 2^60]: g ~ sqrt(x) and h ~ 1/(2 sqrt(x)) from the hardware rsq, then the exact residual
// x - g*g (one FMA) corrects g to the correctly rounded root.  Exhaustively equal to sqrtf(x).
__device__ __forceinline__ float sqrt_fast(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// x in [2^-60, 2^60] as one unsigned compare on the bits (biased exponents 67..187; the upper bound is exactly 2^60).
// Negative values, zeros, subnormals, infinities and NaNs all fail it.
__device__ __forceinline__ bool in_safe_range(float x) { return (__float_as_uint(x) - 0x21800000u) <= (0x5d800000u - 0x21800000u); }

// The short sequence runs on every lane; only when some lane of the wave is outside the safe range (practically
// never on this path) does the wave branch to the generic expansion for those lanes.  One wave-uniform branch that
// is not taken costs two instructions, where a per-lane if/else around five instructions costs ten.
__device__ __forceinline__ float rcp_cr(float x)
{
    float r = rcp_fast(x);
    const bool odd = !in_safe_range(fabsf(x));
    if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
        if (odd) r = rcp_generic(x);
    }
    return r;
}

__device__ __forceinline__ float sqrt_cr(float x)
{
    float r = sqrt_fast(x);
    const bool odd = !in_safe_range(x);
    if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
        if (odd) r = sqrt_generic(x);
    }
    return r;
}

// 1.0f / sqrt(x) as two correctly rounded operations; one range test covers both
// (x in [2^-60, 2^60] puts the root in [2^-30, 2^30]).
__device__ __forceinline__ float rcp_sqrt_cr(float x)
{
    float r = rcp_fast(sqrt_fast(x));
    const bool odd = !in_safe_range(x);
    if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
        if (odd) r = rcp_generic(sqrt_generic(x));
    }
    return r;
}

// MyMath.h:139-145: one reciprocal, three multiplies, no zero check
__device__ __forceinline__ V3 normalize_gpu(V3 a)
{
    const float length = rcp_sqrt_cr(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3(a.x * length, a.y * length, a.z * length);
}
// MyMath.cu:29-34
__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    const float r = v < lo ? lo : v;
    return r > hi ? hi : r;
}
// MyMath.cu:59-62
__device__ __forceinline__ float minf(float a, float b) { return a < b ? a : b; }

// pow(x, 32.0f) of RayTracing.cu:73, pinned: five squarings in double, rounded once to float.
// libm powf is not reproducible across libms; this routine is IEEE-exact on
// CPU and GPU alike and is what the oracle evaluates by default.
__device__ __forceinline__ float pow32(float x)
{
    double d = (double)x;
    d = d * d;
    d = d * d;
    d = d * d;
    d = d * d;
    d = d * d;
    return (float)d;
}

// (uint8_t)f as the CUDA hardware conversion does it: truncate; negatives and NaN give 0, values past
// the u32 range saturate (low byte 255).  v_cvt_u32_f32 has exactly these semantics; it is spelled in
// asm because a C++ float->unsigned cast of an out-of-range value is undefined.
__device__ __forceinline__ uint32_t u8_sat(float f)
{
    uint32_t u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(f));
    return u & 255u;
}

struct Ray {
    V3 o, d;
    float a, fourA, divTwoA;
};

struct Camera {
    float m[12]; // first three rows of inverseVMatrix, row-major
    float ox, oy, oz;
    float e1, e2, far;
    float fW, fH; // (float)W, (float)H
};

// Sphere::Trace, Sphere.cu:30-68, split in two.
//
// sphere_reject: the miss test on the hoisted terms otc = o - c and cc = Dot(otc,otc) - r*r
// (ray-independent for primary rays, which all share the origin).  The reference evaluates
// disc = b*b - fourA*cc with b = 2*s, fourA = 4*a.  Scaling by powers of two commutes with rounding while
// nothing overflows or goes subnormal, so there disc == 4 * (s*s - a*cc) bit for bit and the sign test
// runs on q = s*s - a*cc, one multiply fewer.  The rejection is only taken when q < -1e-30: then at least one
// of the two products is a normal number of that size, a subnormal partner moves neither sum by more than
// 1e-44, and disc is negative too; the sliver -1e-30 <= q < 0 goes on to sphere_hit, which evaluates the literal
// discriminant.  Overflow (|s| > 9.2e18, or a*cc near 1e38) makes disc +-inf or NaN where q is finite, but every
// such case ends in a miss on both routes: disc = -inf is a reject, +inf gives t2 = -inf, NaN gives NaN roots
// that no comparison accepts -- and whenever q does not reject, sphere_hit runs the reference's own arithmetic.
// So the outcome equals the reference's for every input (tests: scenes scaled by 1e18 and 1e-18).
__device__ __forceinline__ bool sphere_reject(const Ray& r, float ox, float oy, float oz, float cc, float& s)
{
    s = r.d.x * ox + r.d.y * oy + r.d.z * oz;
    const float q = s * s - r.a * cc;
    return q < -1.0e-30f;
}

// sphere_hit: the literal reference arithmetic from the discriminant on; true with t set on a hit.
__device__ __forceinline__ bool sphere_hit(const Ray& r, float s, float cc, float& t)
{
    const float b = 2.0f * s;
    const float discriminant = b * b - r.fourA * cc;
    if (discriminant < 0.0f) {
        return false;
    }
    const float sqrtDiscriminant = sqrt_cr(discriminant);
    const float minusB = -b;
    // The reference forms both roots, t1 = (-b + sqrt) * divTwoA and t2 = (-b - sqrt) * divTwoA, misses if either is negative
    // and takes Min(t1, t2) (Sphere.cu:52-66).  Only t2 is needed: sqrt >= +0 and divTwoA = 1 / (2 a) >= +0 (a = Dot(d, d) is a
    // sum of squares), and rounded addition and multiplication by a non-negative factor are monotonic, so t2 <= t1 whenever
    // both are numbers: "t1 < 0 || t2 < 0" is "t2 < 0", and MyMath::Min(t1, t2) = (t1 < t2 ? t1 : t2) is t2.  With a NaN
    // anywhere (NaN or infinite discriminant, 0 * inf) every comparison is false on both routes and the value handed on is t2
    // in both: (t1 < NaN ? t1 : NaN) = NaN, and a NaN t1 loses "t1 < t2".  Same result, four instructions fewer per exact test.
    const float t2 = (minusB - sqrtDiscriminant) * r.divTwoA;
    if (t2 < 0.0f) {
        return false;
    }
    t = t2;
    return true;
}

// Plane::Trace, Plane.cu:38-72.
__device__ __forceinline__ bool plane_hit(const Ray& r, V3 p, V3 n, float width, float height, float& t)
{
    const float dn = dot(r.d, n);
    // FloatEquals(dn, 0.0f), MyMath.cu:43-47
    if (dn > 0.0f || fabsf(dn - 0.0f) < 1.1920928955078125e-7f) {
        return false;
    }
    const float t1 = dot(sub(p, r.o), n) / dn;
    if (t1 <= 0.0f) {
        return false;
    }
    const V3 hp = add(r.o, mulf(r.d, t1));
    const float hw = width * 0.5f;
    const float hh = height * 0.5f;
    if ((hp.x <= p.x - hw || hp.x >= p.x + hw) || (hp.z <= p.z - hh || hp.z >= p.z + hh)) {
        return false;
    }
    t = t1;
    return true;
}

// BlinnPhongShading with the call-site constants, RayTracing.cu:41-79 and :143-157.
// od in: object colour / 255.0f (RayTracing.cu:144; the division is done once per object at upload,
// the same IEEE operation on the same operands); out: shaded colour clamped to <= 255.
__device__ __forceinline__ V3 shade(const Ray& r, float distance, V3 normal, V3 od)
{
    const V3 point = add(r.o, mulf(r.d, distance));
    const V3 viewDir = normalize_gpu(mulf(r.d, -1.0f));

    V3 lightDir = sub(v3(1.0f, 50.0f, 0.0f), point);
    float dist = sqrt_cr(lightDir.x * lightDir.x + lightDir.y * lightDir.y + lightDir.z * lightDir.z);
    dist = dist * dist;
    const float divDistance = rcp_cr(dist);
    lightDir = normalize_gpu(lightDir);

    const V3 nn = normalize_gpu(normal);
    const V3 nv = normalize_gpu(viewDir);

    const float diffuseIntensity = clampf(dot(nn, lightDir), 0.0f, 1.0f);
    // lightDiffuseColour (1,1,1) * intensity * 2000 * divDistance, per component
    const float diffuse = ((1.0f * diffuseIntensity) * 2000.0f) * divDistance;

    const V3 h = normalize_gpu(add(lightDir, nv));
    const float specularIntensity = pow32(clampf(dot(nn, h), 0.0f, 1.0f));
    const float specular = ((1.0f * specularIntensity) * 3000.0f) * divDistance;

    // ComponentMul(ambient, od) + ComponentMul(diffuse, od) + ComponentMul(specular, (1,1,1))
    V3 res;
    res.x = 0.2f * od.x + diffuse * od.x + specular * 1.0f;
    res.y = 0.2f * od.y + diffuse * od.y + specular * 1.0f;
    res.z = 0.2f * od.z + diffuse * od.z + specular * 1.0f;
    res = mulf(res, 255.0f);
    return v3(minf(255.0f, res.x), minf(255.0f, res.y), minf(255.0f, res.z));
}

// GetASCIICharacter's index, RayTracing.cu:26-39.  The reference clamps to 68, one past its
// 68-entry table; the build resolves that to the last glyph.
__device__ __forceinline__ int ramp_index(float shadingValue)
{
    int i = (int)ceilf(shadingValue * 67.0f);
    i = i < 1 ? 1 : i;
    return i > 67 ? 67 : i;
}

// xterm-256 mapping of ANSIRGB.h:114-189, written from the algorithm: palette entries are
// computed, the grey lookup (256 bytes, generated on the host by rule) is read from `grey`.
__device__ __forceinline__ uint32_t ansi_distance(uint32_t r1, uint32_t g1, uint32_t b1, uint32_t r2, uint32_t g2, uint32_t b2)
{
    const int32_t r_sum = (int32_t)(r1 + r2);
    const int32_t r = (int32_t)r1 - (int32_t)r2;
    const int32_t g = (int32_t)g1 - (int32_t)g2;
    const int32_t b = (int32_t)b1 - (int32_t)b2;
    return (uint32_t)((1024 + r_sum) * r * r + 2048 * g * g + (1534 - r_sum) * b * b);
}

__device__ __forceinline__ uint32_t cube_level(uint32_t v, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t t4)
{
    return (uint32_t)(v >= t0) + (uint32_t)(v >= t1) + (uint32_t)(v >= t2) + (uint32_t)(v >= t3) + (uint32_t)(v >= t4);
}

__device__ __forceinline__ uint32_t cube_value(uint32_t level) { return level == 0u ? 0u : 55u + 40u * level; }

__device__ __forceinline__ uint32_t palette_grey_value(uint32_t index)
{
    // only indices the grey lookup produces: 16, 59, 102, 145, 188, 231 (cube diagonal) and 232..255
    return index >= 232u ? 8u + 10u * (index - 232u) : cube_value((index - 16u) / 43u);
}

__device__ __forceinline__ uint32_t ansi256_from_rgb(uint32_t r, uint32_t g, uint32_t b, const uint8_t* __restrict__ grey)
{
    if (r == g && g == b) {
        return grey[b];
    }
    const uint32_t lum = (3567664u * r + 11998547u * g + 1211005u * b + (1u << 23)) >> 24;
    const uint32_t grey_index = grey[lum & 255u];
    const uint32_t gv = palette_grey_value(grey_index);
    const uint32_t grey_distance = ansi_distance(r, g, b, gv, gv, gv);
    const uint32_t ir = cube_level(r, 38u, 115u, 155u, 196u, 235u);
    const uint32_t ig = cube_level(g, 36u, 116u, 154u, 195u, 235u);
    const uint32_t ib = cube_level(b, 35u, 115u, 155u, 195u, 235u);
    const uint32_t cube_distance = ansi_distance(r, g, b, cube_value(ir), cube_value(ig), cube_value(ib));
    return cube_distance < grey_distance ? 16u + 36u * ir + 6u * ig + ib : grey_index;
}

} // namespace rtx
