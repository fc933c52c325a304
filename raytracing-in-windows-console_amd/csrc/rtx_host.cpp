// rtx_host.cpp -- host-side input builders of the C ABI: camera parameters and the synthetic
// BASELINE scenes.  Pure fp32 host math, no GPU work; compile with -ffp-contract=off.
#include "../../include/rtx.h"

#include <cmath>
#include <cstring>

extern "C" {

// Camera3D::Init (Camera3D.cpp:8-48: projection scalars, aspect = width / (0.01 * width * height),
// FOV pi/1.5, far 250: Camera3D.h:74-80), Camera3D::Update (:51-98: basis from pitch/yaw, matrix
// rows (right.i, up.i, forward.i, pos.i)), Camera3D::GetInverseVMatrix (:207-376: the sixteen
// six-term cofactor expansions in that term order, then scaling by 1/det) and the params fill of
// Engine3D::Render (Engine3D.cpp:90-97).  m[k] is row k/4, column k%4.
int rtx_camera_params(size_t w, size_t h, const float pos_in[3], const float rot_in[3], rtx_params* out)
{
    if (!out || w == 0 || h == 0) {
        return RTX_ERR_INVALID_ARGUMENT;
    }
    const float zero3[3] = {0.0f, 0.0f, 0.0f};
    const float start_rot[3] = {0.0f, (float)(3.14159265358979323846), 0.0f}; // Camera3D.h:62
    const float* pos = pos_in ? pos_in : zero3;
    const float* rot = rot_in ? rot_in : start_rot;

    const float currentFOV = (float)(3.14159265358979323846) / 1.5f;
    const float width = (float)w, height = (float)h;
    const float aspect = width / (0.01f * width * height);
    const float e = 1.0f / (std::tan(currentFOV / 2.0f));

    const float p = rot[0], y = rot[1];
    const float sp = std::sin(p), cp = std::cos(p), sy = std::sin(y), cy = std::cos(y);
    const float right[3] = {cy, -sp * sy, -cp * sy};
    const float up[3] = {0.0f, cp, -sp};
    const float fwd[3] = {-sy, -sp * cy, -cp * cy};

    float m[16];
    for (int i = 0; i < 3; i++) {
        m[4 * i + 0] = right[i];
        m[4 * i + 1] = up[i];
        m[4 * i + 2] = fwd[i];
        m[4 * i + 3] = pos[i];
    }
    m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;

    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];

    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0f) {
        return RTX_ERR_INVALID_ARGUMENT; // Camera3D.cpp:349-352 asserts
    }
    det = 1.0f / det;
    for (int k = 0; k < 16; k++) {
        out->inv_v[k] = inv[k] * det;
    }
    out->cam_pos[0] = pos[0];
    out->cam_pos[1] = pos[1];
    out->cam_pos[2] = pos[2];
    out->element1 = e / aspect;
    out->element2 = e;
    out->cam_far = 250.0f;
    out->x = w;
    out->y = h;
    return RTX_OK;
}

// SURVEY.md Appendix D.  LCG s = s*1664525 + 1013904223 (mod 2^32), u01 = (float)(s>>8) * 2^-24,
// ur(a,b) = a + (b-a)*u01, all fp32.  Per sphere, in this draw order: d = ur(40,200),
// tx = e1*ur(-.95,.95), ty = e2*ur(-.95,.95), c = 1/sqrtf(1+tx^2+ty^2), centre (d*c*tx, d*c*ty, d*c),
// r = d*k*ur(.5,1)*c*sqrtf(c) with k = sqrtf(1.4*e1*e2/N), colour floorf(ur(1,256)) x3.
int rtx_synth_scene(uint32_t seed, size_t n_spheres, size_t n_planes, float e1, float e2, float* sph, float* pl)
{
    if ((n_spheres && !sph) || (n_planes && !pl) || n_planes > 6) {
        return RTX_ERR_INVALID_ARGUMENT;
    }
    uint32_t s = seed;
    auto u01 = [&s]() -> float {
        s = s * 1664525u + 1013904223u;
        return (float)(s >> 8) * 5.9604644775390625e-08f;
    };
    auto ur = [&u01](float a, float b) -> float { return a + (b - a) * u01(); };

    const float k = std::sqrt(1.4f * e1 * e2 / (float)n_spheres);
    for (size_t i = 0; i < n_spheres; i++) {
        const float d = ur(40.0f, 200.0f);
        const float tx = e1 * ur(-0.95f, 0.95f);
        const float ty = e2 * ur(-0.95f, 0.95f);
        const float c = 1.0f / std::sqrt(1.0f + tx * tx + ty * ty);
        float* o = sph + 7 * i;
        o[0] = d * c * tx;
        o[1] = d * c * ty;
        o[2] = d * c;
        o[3] = d * k * ur(0.5f, 1.0f) * c * std::sqrt(c);
        o[4] = std::floor(ur(1.0f, 256.0f));
        o[5] = std::floor(ur(1.0f, 256.0f));
        o[6] = std::floor(ur(1.0f, 256.0f));
    }
    // planes, in this order: floor, ceiling, back, left, right, patch
    const float W2 = 2.0f * e1 * 250.0f;
    const float planes[6][11] = {
        {0.0f, -30.0f, 125.0f, 0.0f, 1.0f, 0.0f, 100.0f, 100.0f, 100.0f, W2, 250.0f},
        {0.0f, 30.0f, 125.0f, 0.0f, -1.0f, 0.0f, 60.0f, 90.0f, 160.0f, W2, 250.0f},
        {0.0f, 0.0f, 220.0f, 0.0f, 0.0f, -1.0f, 150.0f, 120.0f, 90.0f, W2, 1.0f},
        {-0.8f * e1 * 200.0f, 0.0f, 125.0f, 1.0f, 0.0f, 0.0f, 160.0f, 60.0f, 60.0f, 1.0f, 250.0f},
        {0.8f * e1 * 200.0f, 0.0f, 125.0f, -1.0f, 0.0f, 0.0f, 60.0f, 160.0f, 60.0f, 1.0f, 250.0f},
        {0.0f, -6.0f, 80.0f, 0.0f, 1.0f, 0.0f, 200.0f, 200.0f, 40.0f, e1 * 60.0f, 30.0f},
    };
    for (size_t j = 0; j < n_planes; j++) {
        std::memcpy(pl + 11 * j, planes[j], sizeof planes[j]);
    }
    return RTX_OK;
}

} // extern "C"
