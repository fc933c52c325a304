/*
 * rtx_oracle.c -- CPU ORACLE (test infrastructure; see rtx_oracle.h for the rules).
 *
 * Structure-faithful plain-C restatement of the reference's per-pixel path.  Every
 * function cites the reference lines it follows (under /root/reference/ConsoleProject/).
 * All arithmetic is fp32, evaluated left to right as the reference writes it; build with
 * -ffp-contract=off and never with -ffast-math (SURVEY.md section 7 "Hard parts").
 */
#include "rtx_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- MyMath ---------- */

/* MyMath.h:8-14 */
static orc_vec3 v3(float x, float y, float z) { orc_vec3 r; r.x = x; r.y = y; r.z = z; return r; }
/* MyMath.h:60-63 */
static orc_vec3 v3_sub(orc_vec3 a, orc_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
/* MyMath.h:74-77 */
static orc_vec3 v3_add(orc_vec3 a, orc_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
/* MyMath.h:88-92 */
static orc_vec3 v3_mulf(orc_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
/* MyMath.h:102-106 */
static orc_vec3 v3_divf(orc_vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
/* MyMath.h:139-145: one reciprocal, three multiplies, no zero check */
static orc_vec3 v3_normalize_gpu(orc_vec3 a)
{
    const float length = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3(a.x * length, a.y * length, a.z * length);
}
/* MyMath.h:117-123: the safe host-side normalise used by the Plane constructor */
static orc_vec3 v3_normalize_safe(orc_vec3 a)
{
    const float length = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    const float divider = length < 0.000001f ? 0.0f : 1.0f / length;
    return v3(a.x * divider, a.y * divider, a.z * divider);
}
/* MyMath.h:159-163 */
static float v3_length(orc_vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* MyMath.cu:5-8 */
static float v3_dot(orc_vec3 a, orc_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* MyMath.cu:23-26 */
static orc_vec3 v3_cmul(orc_vec3 a, orc_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* MyMath.cu:29-34 */
static float clampf(float val, float lo, float hi)
{
    const float result = val < lo ? lo : val;
    return result > hi ? hi : result;
}
/* MyMath.cu:36-41 */
static int clampi(int val, int lo, int hi)
{
    const int result = val < lo ? lo : val;
    return result > hi ? hi : result;
}
/* MyMath.cu:43-47 */
static int float_equals(float f1, float f2) { return fabsf(f1 - f2) < FLT_EPSILON; }
/* MyMath.cu:59-62 */
static float minf(float a, float b) { return a < b ? a : b; }

/* pow(x, 32.0f), RayTracing.cu:73.  See enum orc_flags. */
static float pow32(float x, int flags)
{
    if (flags & ORC_POW_LIBM) {
        return powf(x, 32.0f);
    }
    double d = (double)x;
    d = d * d; /* x^2, exact */
    d = d * d; /* x^4 */
    d = d * d; /* x^8 */
    d = d * d; /* x^16 */
    d = d * d; /* x^32 */
    return (float)d;
}

/* ---------------------------------------------------------------- primitives ------ */

/* Object3D.h:16-26 */
typedef struct trace_in {
    orc_vec3 origin, direction;
    float a, fourA, divTwoA;
} trace_in;

/* Object3D.h:28-33 */
typedef struct trace_out {
    int bHit;
    orc_vec3 normal;
    float distance;
} trace_out;

/* Sphere.cu:30-68 */
static void sphere_trace(const orc_object* s, const trace_in* in, trace_out* out)
{
    const orc_vec3 spherePos = s->center;
    const orc_vec3 objectToCam = v3_sub(in->origin, spherePos);

    const float b = 2.0f * v3_dot(in->direction, objectToCam);
    const float c = v3_dot(objectToCam, objectToCam) - (s->radius * s->radius);
    const float discriminant = b * b - in->fourA * c;
    if (discriminant < 0.0f) {
        return;
    }
    const float sqrtDiscriminant = sqrtf(discriminant);
    const float minusB = -b;
    float t1 = (minusB + sqrtDiscriminant) * in->divTwoA;
    const float t2 = (minusB - sqrtDiscriminant) * in->divTwoA;
    if (t1 < 0.0f || t2 < 0.0f) {
        return;
    }
    t1 = minf(t1, t2);
    out->bHit = 1;
    out->distance = t1;
    out->normal = v3_normalize_gpu(v3_sub(v3_add(in->origin, v3_mulf(in->direction, t1)), spherePos));
}

/* Plane.cu:38-72 */
static void plane_trace(const orc_object* pl, const trace_in* in, trace_out* out)
{
    const orc_vec3 planeNormal = pl->normal;
    const orc_vec3 planePos = pl->center;
    const float dotLineAndPlaneNormal = v3_dot(in->direction, planeNormal);
    if (dotLineAndPlaneNormal > 0.0f || float_equals(dotLineAndPlaneNormal, 0.0f)) {
        return;
    }
    const float t1 = v3_dot(v3_sub(planePos, in->origin), planeNormal) / dotLineAndPlaneNormal;
    if (t1 <= 0.0f) {
        return;
    }
    const orc_vec3 hitPoint = v3_add(in->origin, v3_mulf(in->direction, t1));
    const float halfPlaneWidth = pl->width * 0.5f;
    const float halfPlaneHeight = pl->height * 0.5f;
    if ((hitPoint.x <= planePos.x - halfPlaneWidth || hitPoint.x >= planePos.x + halfPlaneWidth) ||
        (hitPoint.z <= planePos.z - halfPlaneHeight || hitPoint.z >= planePos.z + halfPlaneHeight)) {
        return;
    }
    out->bHit = 1;
    out->distance = t1;
    out->normal = planeNormal;
}

/* ---------------------------------------------------------------- RayTracing.cu --- */

/* RayTracing.h:97-115.  Index 68 is one past the reference's table (RayTracing.cu:36 clamps to
 * NUM_ASCII_CHARACTERS, not NUM_ASCII_CHARACTERS-1); the build resolves that read to '@'
 * (SURVEY App. E-3), so the table here carries a 69th entry equal to the 68th. */
static const char ORC_RAMP[69] = {
    ' ', '.', '`', '^', '"', ',', ':', ';', 'I', 'l', '!', 'i', '>', '<', '~', '+', '_',
    '-', '?', '*', ']', '[', '}', '{', '1', ')', '(', '|', '/', 't', 'f', 'j', 'r', 'x',
    'n', 'u', 'v', 'c', 'z', 'm', 'w', 'X', 'Y', 'U', 'J', 'C', 'L', 'q', 'p', 'd', 'b',
    'k', 'h', 'a', 'o', '#', '%', 'Z', 'O', '8', 'B', '$', '0', 'Q', 'M', '&', 'W', '@',
    '@'
};

/* RayTracing.cu:9-24 */
static orc_vec3 initial_direction(const orc_params* p, size_t row, size_t column)
{
    const float convertedY = ((float)(p->y) - (float)(row * 2)) / (float)(p->y);
    const float convertedX = ((float)(2 * column) - (float)(p->x)) / (float)(p->x);

    const float vx = convertedX * p->element1;
    const float vy = convertedY * p->element2;
    const float vz = 1.0f;
    const float vw = 0.0f;

    /* MyMath.h:310-319, first three rows */
    orc_vec3 d;
    d.x = p->inv_v[0][0] * vx + p->inv_v[0][1] * vy + p->inv_v[0][2] * vz + p->inv_v[0][3] * vw;
    d.y = p->inv_v[1][0] * vx + p->inv_v[1][1] * vy + p->inv_v[1][2] * vz + p->inv_v[1][3] * vw;
    d.z = p->inv_v[2][0] * vx + p->inv_v[2][1] * vy + p->inv_v[2][2] * vz + p->inv_v[2][3] * vw;
    return v3_normalize_gpu(d);
}

/* RayTracing.cu:26-39; returns the ramp index (0 = blank) */
static int ramp_index(float distance, float farPlane, float shadingValue)
{
    if (distance > farPlane) {
        return 0;
    }
    return clampi((int)ceilf(shadingValue * (float)(68 - 1)), 1, 68);
}

/* RayTracing.cu:41-79 */
static orc_vec3 blinn_phong(orc_vec3 objectDiffuseColour, orc_vec3 objectSpecularColour, orc_vec3 lightPos,
                            orc_vec3 lightDiffuseColour, float lightDiffusePower,
                            orc_vec3 lightSpecularColour, float lightSpecularPower,
                            orc_vec3 point, orc_vec3 viewDir, orc_vec3 normal, int flags)
{
    orc_vec3 lightDir = v3_sub(lightPos, point);

    float distance = v3_length(lightDir);
    distance = distance * distance;
    const float divDistance = 1.0f / distance;

    lightDir = v3_normalize_gpu(lightDir);

    const orc_vec3 normalizedNormal = v3_normalize_gpu(normal);
    const orc_vec3 normalizedViewDir = v3_normalize_gpu(viewDir);

    const float NdotL = v3_dot(normalizedNormal, lightDir);
    const float diffuseIntensity = clampf(NdotL, 0.0f, 1.0f);

    const orc_vec3 diffuse =
        v3_mulf(v3_mulf(v3_mulf(lightDiffuseColour, diffuseIntensity), lightDiffusePower), divDistance);

    const orc_vec3 h = v3_normalize_gpu(v3_add(lightDir, normalizedViewDir));

    const float NdotH = v3_dot(normalizedNormal, h);
    const float specularIntensity = pow32(clampf(NdotH, 0.0f, 1.0f), flags);

    const orc_vec3 specular =
        v3_mulf(v3_mulf(v3_mulf(lightSpecularColour, specularIntensity), lightSpecularPower), divDistance);

    const orc_vec3 ambientLight = v3(0.2f, 0.2f, 0.2f);
    return v3_add(v3_add(v3_cmul(ambientLight, objectDiffuseColour), v3_cmul(diffuse, objectDiffuseColour)),
                  v3_cmul(specular, objectSpecularColour));
}

/* RayTracing.h:17-23 defaults + RayTracing.cu:81-168 */
static void ray_trace(orc_vec3 origin, orc_vec3 direction, const orc_object* const* objects, unsigned count,
                      int flags, orc_pixel* ret)
{
    trace_in in;
    in.origin = origin;
    in.direction = direction;
    in.a = v3_dot(direction, direction);
    in.fourA = 4.0f * in.a;
    in.divTwoA = 1.0f / (2.0f * in.a);

    /* Not reset between objects, as the reference (RayTracing.cu:95). */
    trace_out out;
    out.bHit = 0;
    out.normal = v3(0.0f, 0.0f, 0.0f);
    out.distance = 99999999.f;

    int bHitSomething = 0;

    for (size_t i = 0; i < count; i++) {
        const orc_object* o = objects[i];
        switch (o->type) {
        case ORC_PLANE:
            plane_trace(o, &in, &out);
            break;
        case ORC_SPHERE:
            sphere_trace(o, &in, &out);
            break;
        default:
            break;
        }

        if (out.bHit && out.distance < ret->distance) {
            bHitSomething = 1;
            ret->distance = out.distance;
            ret->normal = out.normal;
            ret->normal = v3_normalize_gpu(ret->normal);
            ret->shading_value = v3_dot(ret->normal, v3(1.0f, 0.0f, 0.0f));
            ret->color = o->color;
        }
    }

    ret->hit = bHitSomething;
    if (!bHitSomething) {
        return;
    }

    orc_vec3 shading = blinn_phong(
        v3_divf(ret->color, 255.0f),
        v3(1.0f, 1.0f, 1.0f),
        v3(1.0f, 50.0f, 0.0f),
        v3(1.0f, 1.0f, 1.0f), 2000.0f,
        v3(1.0f, 1.0f, 1.0f), 3000.0f,
        v3_add(origin, v3_mulf(direction, ret->distance)),
        v3_normalize_gpu(v3_mulf(direction, -1.0f)),
        ret->normal, flags);

    shading = v3_mulf(shading, 255.0f);
    ret->color = v3(minf(255.0f, shading.x), minf(255.0f, shading.y), minf(255.0f, shading.z));
}

/* (uint8_t)f as the CUDA hardware conversion does it: truncate, negatives and NaN to 0. */
static uint8_t u8_sat(float f)
{
    if (!(f > 0.0f)) {
        return 0;
    }
    if (f >= 4294967296.0f) {
        return 255; /* u32 saturates to 0xffffffff; its low byte */
    }
    return (uint8_t)(uint32_t)f;
}

/* (uint8_t)f as x86-64 gcc does it: cvttss2si to int32, keep the low byte. */
static uint8_t u8_wrap(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) {
        return 0; /* the "integer indefinite" 0x80000000 has low byte 0 */
    }
    return (uint8_t)(uint32_t)(int32_t)f;
}

/* The three-digit encoder repeated through RayTracing.cu:212-229, 385-443, 663-724:
 * NUL (not '0') for absent leading digits. */
static void digits3(uint8_t value, char* d)
{
    uint8_t index = value;
    const uint8_t originalIndex = value;
    uint8_t tens = index % 100;
    const uint8_t singles = tens % 10;
    char first = '\0', second = '\0', third;
    if (index >= 100) {
        index = (uint8_t)((index - tens) * 0.01f);
        first = (char)(index + '0');
    }
    if (tens >= 10 || originalIndex >= 100) {
        tens = (uint8_t)((tens - singles) * 0.1f);
        second = (char)(tens + '0');
    }
    third = (char)(singles + '0');
    d[0] = first;
    d[1] = second;
    d[2] = third;
}

/* One thread of RayTrace_ASCII / _PIXEL / _RGB_ASCII / _RGB_PIXEL / _RGB_NORMALS / _SDL
 * (RayTracing.cu:170-253, 255-333, 335-473, 475-610, 612-752, 754-795). */
static void trace_and_encode(const orc_params* p, const orc_object* const* objects, unsigned count, int mode,
                             size_t row, size_t column, int flags, char* result, orc_pixel* px_out)
{
    orc_pixel px;
    px.distance = 99999999.f;
    px.shading_value = 0.0f;
    px.normal = v3(0.0f, 0.0f, 0.0f);
    px.color = v3(0.0f, 0.0f, 0.0f);
    px.hit = 0;
    px.ramp_index = 0;
    px.ansi_index = -1;

    const orc_vec3 dir = initial_direction(p, row, column);
    const orc_vec3 origin = v3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]);
    ray_trace(origin, dir, objects, count, flags, &px);

    const int visible = px.distance <= p->cam_far;
    px.ramp_index = ramp_index(px.distance, p->cam_far, px.shading_value);
    const char data = ORC_RAMP[px.ramp_index];
    if (visible) {
        px.ansi_index = orc_ansi256_from_rgb(((uint32_t)u8_sat(px.color.x) << 16) +
                                             ((uint32_t)u8_sat(px.color.y) << 8) + (uint32_t)u8_sat(px.color.z));
    }
    if (px_out) {
        *px_out = px;
    }
    if (!result || mode == ORC_SDL) {
        return;
    }

    if (mode == ORC_BIT_ASCII || mode == ORC_BIT_PIXEL) {
        char* dst = result + (row * (p->x * 12) + column * 12);
        if (visible) {
            char d[3];
            digits3((uint8_t)px.ansi_index, d);
            const char rec[12] = { '\x1b', '[', mode == ORC_BIT_ASCII ? '3' : '4', '8', ';', '5', ';',
                                   d[0], d[1], d[2], 'm', mode == ORC_BIT_ASCII ? data : ' ' };
            memcpy(dst, rec, 12);
        } else {
            const char rec[12] = { '\x1b', '[', '4', '8', ';', '5', ';', '\0', '1', '6', 'm', ' ' };
            memcpy(dst, rec, 12);
        }
        return;
    }

    char* dst = result + (row * (p->x * 20) + column * 20);
    if (visible) {
        uint8_t r, g, b;
        if (mode == ORC_RGB_NORMALS) {
            /* RayTracing.cu:649 divides colour by shadingValue; the result is never read. */
            if (flags & ORC_NORMALS_WRAP) {
                r = u8_wrap(px.normal.x * 255);
                g = u8_wrap(px.normal.y * 255);
                b = u8_wrap(px.normal.z * 255);
            } else {
                r = u8_sat(px.normal.x * 255);
                g = u8_sat(px.normal.y * 255);
                b = u8_sat(px.normal.z * 255);
            }
        } else {
            r = u8_sat(px.color.x);
            g = u8_sat(px.color.y);
            b = u8_sat(px.color.z);
        }
        char dr[3], dg[3], db[3];
        digits3(r, dr);
        digits3(g, dg);
        digits3(b, db);
        const char rec[20] = { '\x1b', '[', mode == ORC_RGB_ASCII ? '3' : '4', '8', ';', '2', ';',
                               dr[0], dr[1], dr[2], ';', dg[0], dg[1], dg[2], ';', db[0], db[1], db[2],
                               'm', mode == ORC_RGB_ASCII ? data : ' ' };
        memcpy(dst, rec, 20);
    } else {
        const char rec[20] = { '\x1b', '[', '4', '8', ';', '2', ';', '\0', '\0', '0', ';',
                               '\0', '\0', '0', ';', '\0', '\0', '0', 'm', ' ' };
        memcpy(dst, rec, 20);
    }
}

int orc_trace_pixel(const orc_params* p, const orc_object* const* objects, unsigned count,
                    size_t row, size_t col, int flags, orc_pixel* out)
{
    /* RayTracing.cu:187: the last column is reserved for the newline */
    if (p->x == 0 || col >= (p->x - 1) || row >= p->y) {
        return -1;
    }
    trace_and_encode(p, objects, count, ORC_SDL, row, col, flags, NULL, out);
    return 0;
}

int orc_render_rows(const orc_params* p, const orc_object* const* objects, unsigned count, int mode,
                    size_t row0, size_t rows, int flags, char* result, orc_pixel* pixels)
{
    if (mode < ORC_BIT_ASCII || mode > ORC_SDL) {
        return -1; /* RayTracing.cu:863-865 asserts */
    }
    if (p->x == 0) {
        return 0;
    }
    size_t row_end = row0 + rows;
    if (row_end > p->y) {
        row_end = p->y;
    }
    for (size_t row = row0; row < row_end; row++) {
        for (size_t col = 0; col + 1 < p->x; col++) {
            trace_and_encode(p, objects, count, mode, row, col, flags, result,
                             pixels ? pixels + row * p->x + col : NULL);
        }
    }
    return 0;
}

/* One row of the frame into a buffer of its own (x * S bytes, S = 12 or 20 by mode; zero-initialised by the caller):
 * what a checker needs when the frame is 8K and only a few rows are to be compared. */
int orc_render_row(const orc_params* p, const orc_object* const* objects, unsigned count, int mode,
                   size_t row, int flags, char* row_out)
{
    const size_t S = (mode == ORC_BIT_ASCII || mode == ORC_BIT_PIXEL) ? 12u : 20u;
    /* orc_render_rows addresses the record of (row, col) at result + row * x * S + col * S: hand it the address
     * that puts this row's first record at row_out (integer arithmetic: no pointer outside an object is formed) */
    char* base = (char*)((uintptr_t)row_out - (uintptr_t)(row * p->x * S));
    return orc_render_rows(p, objects, count, mode, row, 1, flags, base, NULL);
}

typedef struct mt_shared {
    const orc_params* p;
    const orc_object* const* objects;
    unsigned count;
    int mode, flags;
    size_t next_row; /* first row nobody has taken yet (atomic) */
    char* result;
} mt_shared;

/* rows a worker takes at a time: small enough that the last blocks end together whatever the scene's
 * density per row, large enough that the shared counter is touched once per ~10^4 pixels */
#define ORC_MT_BLOCK_ROWS 4

static void* mt_worker(void* arg)
{
    mt_shared* j = (mt_shared*)arg;
    const size_t H = j->p->y;
    for (;;) {
        const size_t row0 = __atomic_fetch_add(&j->next_row, (size_t)ORC_MT_BLOCK_ROWS, __ATOMIC_RELAXED);
        if (row0 >= H) {
            break;
        }
        orc_render_rows(j->p, j->objects, j->count, j->mode, row0, ORC_MT_BLOCK_ROWS, j->flags, j->result, NULL);
    }
    return NULL;
}

int orc_render_mt(const orc_params* p, const orc_object* const* objects, unsigned count, int mode,
                  int flags, int nthreads, char* result)
{
    if (mode < ORC_BIT_ASCII || mode > ORC_SDL) {
        return -1; /* RayTracing.cu:863-865 asserts */
    }
    if (nthreads < 1) {
        nthreads = 1;
    }
    if (nthreads > 1024) {
        nthreads = 1024;
    }
    /* Row blocks ("row-block partition"), handed out dynamically: every worker draws the next
     * block of ORC_MT_BLOCK_ROWS rows from a shared counter, so that rows over dense parts of the scene do not
     * leave the other threads idle at the end (a static block per thread did: VERDICT r03).  Pixels are
     * independent, so the frame is the same bytes whatever the schedule. */
    static pthread_t tid[1024];
    static char joined[1024];
    mt_shared job;
    job.p = p;
    job.objects = objects;
    job.count = count;
    job.mode = mode;
    job.flags = flags;
    job.next_row = 0;
    job.result = result;
    for (int t = 0; t + 1 < nthreads; t++) {
        joined[t] = pthread_create(&tid[t], NULL, mt_worker, &job) == 0 ? 1 : 0;
    }
    mt_worker(&job); /* the caller is the last of the `nthreads` workers (and works alone if no thread could be started) */
    for (int t = 0; t + 1 < nthreads; t++) {
        if (joined[t]) {
            pthread_join(tid[t], NULL);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- Minimize --------- */

/* RayTracingManager.cu:181-249 (S = 12, 3 colour bytes at +7) and :251-319 (S = 20, 9 colour
 * bytes at +7..+9, +11..+13, +15..+17).  The RGB variant stops on PrintMachine::GetHeight()
 * (:306), which equals y on the path (Engine3D.cpp:93). */
size_t orc_minimize(int mode, const char* in, size_t size, size_t x, size_t y, char* out)
{
    const int rgb = !(mode == ORC_BIT_ASCII || mode == ORC_BIT_PIXEL);
    const size_t S = rgb ? 20 : 12;
    size_t newlines = 0;
    size_t addedChars = 0;
    const char* latestColor = NULL;

    for (size_t i = 0; i < size;) {
        const char current = in[i];
        if (current == '\x1b') {
            int differs;
            if (!latestColor) {
                differs = 1;
            } else if (!rgb) {
                differs = latestColor[0] != in[i + 7] || latestColor[1] != in[i + 8] || latestColor[2] != in[i + 9];
            } else {
                differs = latestColor[0] != in[i + 7] || latestColor[1] != in[i + 8] || latestColor[2] != in[i + 9] ||
                          latestColor[4] != in[i + 11] || latestColor[5] != in[i + 12] || latestColor[6] != in[i + 13] ||
                          latestColor[8] != in[i + 15] || latestColor[9] != in[i + 16] || latestColor[10] != in[i + 17];
            }
            if (differs) {
                latestColor = in + i + 7;
                memcpy(out + addedChars, in + i, S);
                addedChars += S;
            } else {
                out[addedChars] = in[i + S - 1];
                addedChars += 1;
            }
            i += S;
        } else if (((i + 1) % (S * x)) == 0) {
            ++newlines;
            out[addedChars] = '\n';
            ++addedChars;
            ++i;
            if (newlines == y) {
                break;
            }
        } else {
            ++i;
        }
    }
    return addedChars;
}

/* ---------------------------------------------------------------- ANSIRGB ---------- */
/*
 * Independent implementation of the algorithm ANSIRGB.h:114-189 describes (that header is
 * LGPL-encumbered, so nothing is pasted from it): the xterm-256 palette and the grey lookup
 * are generated from their definitions; tests/test_oracle_pins.py checks the exhaustive
 * 2^24-input hash SURVEY 8(c) recorded.
 */
static uint32_t g_palette[256];
static uint8_t g_grey[256];
static pthread_once_t g_ansi_once = PTHREAD_ONCE_INIT;

static void ansi_init(void)
{
    static const uint8_t level[6] = { 0, 95, 135, 175, 215, 255 };
    /* 0..15: xterm's default system colours.  Never consulted by the mapper. */
    static const uint32_t sys[16] = { 0x000000, 0xcd0000, 0x00cd00, 0xcdcd00, 0x0000ee, 0xcd00cd, 0x00cdcd, 0xe5e5e5,
                                      0x7f7f7f, 0xff0000, 0x00ff00, 0xffff00, 0x5c5cff, 0xff00ff, 0x00ffff, 0xffffff };
    for (int i = 0; i < 16; i++) {
        g_palette[i] = sys[i];
    }
    for (int r = 0; r < 6; r++) {
        for (int g = 0; g < 6; g++) {
            for (int b = 0; b < 6; b++) {
                g_palette[16 + 36 * r + 6 * g + b] = ((uint32_t)level[r] << 16) | ((uint32_t)level[g] << 8) | level[b];
            }
        }
    }
    for (int i = 0; i < 24; i++) {
        const uint32_t v = (uint32_t)(8 + 10 * i);
        g_palette[232 + i] = (v << 16) | (v << 8) | v;
    }
    /* Grey lookup: nearest of the 30 grey palette entries (the cube diagonal 16, 59, 102,
     * 145, 188, 231 and the ramp 232..255).  Exact ties (v midway between two ramp entries)
     * go to the darker entry for v < 120 and to the lighter one above, which is how the
     * reference's table resolves them (ANSIRGB.h:143-176, checked by the exhaustive hash). */
    for (int v = 0; v < 256; v++) {
        const int prefer_lighter = v >= 120;
        int best_idx = 16, best_val = 0, best_d = v;
        for (int k = 1; k < 30; k++) {
            const int val = k < 6 ? level[k] : 8 + 10 * (k - 6);
            const int idx = k < 6 ? 16 + 43 * k : 232 + (k - 6);
            const int d = abs(v - val);
            if (d < best_d || (d == best_d && (prefer_lighter ? val > best_val : val < best_val))) {
                best_d = d; best_val = val; best_idx = idx;
            }
        }
        g_grey[v] = (uint8_t)best_idx;
    }
}

/* ANSIRGB.h:118-124: red-mean weighted distance, u32 arithmetic */
static uint32_t ansi_distance(uint32_t x, uint32_t y)
{
    const int32_t r_sum = (int32_t)((x >> 16) & 0xff) + (int32_t)((y >> 16) & 0xff);
    const int32_t r = (int32_t)((x >> 16) & 0xff) - (int32_t)((y >> 16) & 0xff);
    const int32_t g = (int32_t)((x >> 8) & 0xff) - (int32_t)((y >> 8) & 0xff);
    const int32_t b = (int32_t)(x & 0xff) - (int32_t)(y & 0xff);
    return (uint32_t)((1024 + r_sum) * r * r + 2048 * g * g + (1534 - r_sum) * b * b);
}

/* ANSIRGB.h:8-34: per-channel cube level from thresholds */
static int cube_level(uint8_t v, int t0, int t1, int t2, int t3, int t4)
{
    if (v < t0) return 0;
    if (v < t1) return 1;
    if (v < t2) return 2;
    if (v < t3) return 3;
    if (v < t4) return 4;
    return 5;
}

/* ANSIRGB.h:141-189 */
uint8_t orc_ansi256_from_rgb(uint32_t rgb)
{
    pthread_once(&g_ansi_once, ansi_init);
    const uint8_t r = (rgb >> 16) & 0xff, g = (rgb >> 8) & 0xff, b = rgb & 0xff;
    if (r == g && g == b) {
        return g_grey[b];
    }
    /* ANSIRGB.h:126-139: fixed-point luminance */
    const uint32_t lum = (3567664u * r + 11998547u * g + 1211005u * b + (1u << 23)) >> 24;
    const uint8_t grey_index = g_grey[lum & 0xff];
    const uint32_t grey_distance = ansi_distance(rgb, g_palette[grey_index]);
    const int ir = cube_level(r, 38, 115, 155, 196, 235);
    const int ig = cube_level(g, 36, 116, 154, 195, 235);
    const int ib = cube_level(b, 35, 115, 155, 195, 235);
    const uint8_t cube_index = (uint8_t)(16 + 36 * ir + 6 * ig + ib);
    return ansi_distance(rgb, g_palette[cube_index]) < grey_distance ? cube_index : grey_index;
}

uint64_t orc_fnv1a64_from(const void* data, size_t n, uint64_t h)
{
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

uint64_t orc_fnv1a64(const void* data, size_t n) { return orc_fnv1a64_from(data, n, ORC_FNV_OFFSET_STANDARD); }

uint64_t orc_ansi256_exhaustive_hash(uint64_t h)
{
    for (uint32_t rgb = 0; rgb < (1u << 24); rgb++) {
        h ^= orc_ansi256_from_rgb(rgb);
        h *= 0x100000001b3ull;
    }
    return h;
}

/* out[i] = orc_ansi256_from_rgb(first + i), i < count: the byte sequence the device mapper is compared with */
void orc_ansi256_fill(uint32_t first, size_t count, uint8_t* out)
{
    for (size_t i = 0; i < count; i++) {
        out[i] = orc_ansi256_from_rgb(first + (uint32_t)i);
    }
}

/* ---------------------------------------------------------------- Camera3D --------- */

/* Camera3D.cpp:8-48 (projection scalars), :51-98 (basis + "view" matrix), :207-376 (cofactor
 * inverse, the classic sixteen six-term expansions with m[k] = row k/4, column k%4),
 * Engine3D.cpp:90-97 (params fill).  FOV pi/1.5, far 250: Camera3D.h:74-80. */
void orc_camera_params(size_t w, size_t h, const float pos[3], const float rot[3], orc_params* out)
{
    const float fovDiv = 1.5f, screenFar = 250.0f;
    const float currentFOV = (float)(3.14159265358979323846) / fovDiv;
    const float width = (float)w, height = (float)h;
    const float aspect = width / (0.01f * width * height);
    const float e = 1.0f / (tanf(currentFOV / 2.0f));

    const float p = rot[0], y = rot[1];
    const float fx = -sinf(y), fy = -sinf(p) * cosf(y), fz = -cosf(p) * cosf(y);
    const float rx = cosf(y), ry = -sinf(p) * sinf(y), rz = -cosf(p) * sinf(y);
    const float ux = 0.0f, uy = cosf(p), uz = -sinf(p);

    const float m[16] = { rx, ux, fx, pos[0],
                          ry, uy, fy, pos[1],
                          rz, uz, fz, pos[2],
                          0.0f, 0.0f, 0.0f, 1.0f };
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
    det = 1.0f / det;
    for (int k = 0; k < 16; k++) {
        out->inv_v[k / 4][k % 4] = inv[k] * det;
    }
    out->cam_pos[0] = pos[0];
    out->cam_pos[1] = pos[1];
    out->cam_pos[2] = pos[2];
    out->x = w;
    out->y = h;
    out->element1 = e / aspect; /* pMatrix.row1.x */
    out->element2 = e;          /* pMatrix.row2.y */
    out->cam_far = screenFar;
}

/* ---------------------------------------------------------------- UpdateObjects ---- */

/* RayTracingManager.cu:10-44, Sphere.cu:15-23 (long double is double in device code),
 * Plane.cu:14-18 (no-op). */
void orc_update_objects(orc_object* const* objects, unsigned count, double dt)
{
    for (unsigned i = 0; i < count; i++) {
        orc_object* o = objects[i];
        if (o->type != ORC_SPHERE) {
            continue;
        }
        o->center.y = (float)((double)o->center.y + (double)(o->speed * (float)o->mover) * dt);
        if (o->center.y < -10.0f || o->center.y > 10.0f) {
            o->center.y = clampf(o->center.y, -10.0f, 10.0f);
            o->mover *= -1;
        }
    }
}

/* Exposed for scene construction in tests: Plane.cu:6-12 normalises the normal on the host. */
orc_vec3 orc_plane_normal(orc_vec3 n) { return v3_normalize_safe(n); }
