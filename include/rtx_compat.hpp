// rtx_compat.hpp -- the reference's C++ API surface for the ray-trace path, rebuilt over the C ABI.
//
// Header-only host code.  It keeps the names, signatures, argument meaning and call order of the
// reference's classes so that Engine3D-style plumbing (Engine3D.cpp:9-28, 81-107) compiles against
// it unchanged, while every device-side action goes through include/rtx.h into librtx_hip.so:
//
//   MyMath::Vector3 / Vector4 / Matrix      MyMath.h:5-321   (PODs here: no vptrs travel to the GPU)
//   DeviceObjectArray<T>, Object3D          Object3D.h:6-12, 36-65 (opaque handle here)
//   RayTracingCPUToGPUData, RenderingMode   RayTracingManager.h:9-21
//   PrintMachine                            PrintMachine.h:16-45: Start, GetMaxSize, GetWidth, GetHeight,
//                                           SetDataInBackBuffer, GetBackBuffer, GetPrintSize; the printer thread
//                                           (PrintMachine.cpp:257-306) for a POSIX terminal via StartPrinter(fd)
//   Scene3D                                 Scene3D.h:15-25
//   Camera3D                                Camera3D.h:12-48 (Init, Update, Move, AddRot, m_Keys, SetRot, SetPos, getters)
//   RayTracingManager                       RayTracingManager.h:27-37
//   RayTracing::RayTrace                    RayTracing.h:31-38
//
// Differences, all deliberate: errors throw std::runtime_error with the library's message instead of
// exit() (pch.h:45-53); Scene3D has no 5 MB arena cap (Scene3D.h:6); Sphere speed is not drawn from
// rand() (Sphere.cu:11-12) but defaults to 1 and can be set; RayTracing::RayTrace takes the params as a
// HOST pointer (they travel as kernel arguments) and ignores gridDims/blockDims (tiling is chosen
// for gfx950 inside the library).
#pragma once

#include "rtx.h"

#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace MyMath {

struct Vector3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Vector3() = default;
    Vector3(float inX, float inY, float inZ) : x(inX), y(inY), z(inZ) {}
    Vector3(int inX, int inY, int inZ) : x((float)inX), y((float)inY), z((float)inZ) {}
    Vector3 operator-(const Vector3& o) const { return Vector3(x - o.x, y - o.y, z - o.z); }
    Vector3 operator+(const Vector3& o) const { return Vector3(x + o.x, y + o.y, z + o.z); }
    Vector3 operator*(float s) const { return Vector3(x * s, y * s, z * s); }
    Vector3 operator/(float s) const { return Vector3(x / s, y / s, z / s); }
};

struct Vector4 {
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
    Vector4() = default;
    Vector4(float inX, float inY, float inZ, float inW) : x(inX), y(inY), z(inZ), w(inW) {}
    Vector4(const Vector3& v, float inW) : x(v.x), y(v.y), z(v.z), w(inW) {}
    Vector3 xyz() const { return Vector3(x, y, z); }
};

struct Matrix {
    Vector4 row1, row2, row3, row4;
    Matrix() = default;
    Matrix(const Vector4& a, const Vector4& b, const Vector4& c, const Vector4& d) : row1(a), row2(b), row3(c), row4(d) {}
    Vector4 Mult(const Vector4& v) const
    {
        return Vector4(row1.x * v.x + row1.y * v.y + row1.z * v.z + row1.w * v.w,
                       row2.x * v.x + row2.y * v.y + row2.z * v.z + row2.w * v.w,
                       row3.x * v.x + row3.y * v.y + row3.z * v.z + row3.w * v.w,
                       row4.x * v.x + row4.y * v.y + row4.z * v.z + row4.w * v.w);
    }
};

inline float Dot(const Vector3& a, const Vector3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

} // namespace MyMath

// ---------------------------------------------------------------------------------------------

class Object3D; // opaque: objects live in the library's SoA store

template <typename T>
struct DeviceObjectArray {
    T* m_deviceArray = nullptr; // here: an opaque handle to the scene store of the shared context
    unsigned int allocatedBytes = 0;
    unsigned int count = 0;
};

struct RayTracingCPUToGPUData {
    MyMath::Matrix inverseVMatrix;
    MyMath::Vector3 camPos;
    size_t x = 0;
    size_t y = 0;
    float element1 = 0.0f;
    float element2 = 0.0f;
    float camFarDist = 0.0f;
};

enum RenderingMode { BIT_ASCII = 0, BIT_PIXEL, RGB_ASCII, RGB_PIXEL, RGB_NORMALS, SDL };

struct dim3_compat {
    unsigned x = 1, y = 1, z = 1;
    dim3_compat() = default;
    dim3_compat(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace rtx_compat {

inline rtx_params to_rtx_params(const RayTracingCPUToGPUData& p)
{
    rtx_params r;
    const MyMath::Vector4* rows[4] = {&p.inverseVMatrix.row1, &p.inverseVMatrix.row2, &p.inverseVMatrix.row3, &p.inverseVMatrix.row4};
    for (int i = 0; i < 4; i++) {
        r.inv_v[4 * i + 0] = rows[i]->x;
        r.inv_v[4 * i + 1] = rows[i]->y;
        r.inv_v[4 * i + 2] = rows[i]->z;
        r.inv_v[4 * i + 3] = rows[i]->w;
    }
    r.cam_pos[0] = p.camPos.x;
    r.cam_pos[1] = p.camPos.y;
    r.cam_pos[2] = p.camPos.z;
    r.element1 = p.element1;
    r.element2 = p.element2;
    r.cam_far = p.camFarDist;
    r.x = p.x;
    r.y = p.y;
    return r;
}

inline void check(rtx_ctx* ctx, int status, const char* what)
{
    if (status != RTX_OK) {
        throw std::runtime_error(std::string(what) + ": " + rtx_last_error(ctx));
    }
}

// One device context per process, like the reference's single default device -- or one device GROUP: with a
// device list (Device::set_devices({0, 1, 2, 3}) before the first use, or RTX_DEVICES=0,1,2,3 in the environment) the same
// context renders every frame row-sharded over those GPUs and assembles it on the first (rtx_group_create, rtx.h), and
// Engine3D::Render-style code -- Scene3D::CreateSphere, RayTracingManager::Update -- runs unchanged on top.
struct Device {
    static rtx_ctx*& slot()
    {
        static rtx_ctx* ctx = nullptr;
        return ctx;
    }
    static int& ordinal()
    {
        static int dev = 0;
        return dev;
    }
    static std::vector<int>& devices()
    {
        static std::vector<int> list;
        return list;
    }
    static void set_devices(const std::vector<int>& list) { devices() = list; }
    static std::vector<int> devices_from_environment()
    {
        std::vector<int> list;
        const char* env = std::getenv("RTX_DEVICES");
        if (!env) return list;
        const char* p = env;
        while (*p) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            list.push_back((int)v);
            p = end;
            while (*p == ',' || *p == ' ') p++;
        }
        return list;
    }
    static rtx_ctx* get(size_t max_w, size_t max_h)
    {
        rtx_ctx*& ctx = slot();
        if (!ctx) {
            std::vector<int> list = devices().empty() ? devices_from_environment() : devices();
            const int rc = list.empty() ? rtx_create(ordinal(), max_w, max_h, &ctx)
                                        : rtx_group_create((int)list.size(), list.data(), max_w, max_h, &ctx);
            if (rc != RTX_OK) {
                throw std::runtime_error(std::string(list.empty() ? "rtx_create: " : "rtx_group_create: ") + rtx_last_error(nullptr));
            }
        }
        return ctx;
    }
    static void release()
    {
        if (slot()) {
            rtx_destroy(slot());
            slot() = nullptr;
        }
    }
};

} // namespace rtx_compat

// ---------------------------------------------------------------------------------------------
// PrintMachine: the statics the hot path calls (RayTracingManager.cu:58,150,306) with the double buffer and mutex of
// PrintMachine.cpp:178-192, and -- on request -- the printer thread of PrintMachine.cpp:257-306 for a POSIX
// terminal: StartPrinter(fd) spawns it; it swaps the back buffer in under the mutex, homes the cursor with an ANSI
// escape (the reference calls SetConsoleCursorPosition(0,0)) and writes the frame to fd.  Without StartPrinter the
// class is headless (the caller reads GetBackBuffer / GetPrintSize itself).  One deliberate difference: the
// reference's thread re-prints the same buffer in a busy loop between frames; this one sleeps on a condition
// variable until SetDataInBackBuffer flags a new frame (a terminal gains nothing from identical bytes), and a frame
// that arrives while the previous one is still being written replaces it, as in the reference.
class PrintMachine {
public:
    static void Start(const size_t x, const size_t y)
    {
        State& s = state();
        s.width = x;
        s.height = y;
        s.maxSize = 20 * x * y; // m_charsPerPixel, PrintMachine.h:81
        s.backBuffer.assign(s.maxSize, 0);
        s.printBuffer.assign(s.maxSize, 0);
        s.backBufferPrintSize = 0;
        s.shouldSwap = false;
    }
    static void CleanUp()
    {
        StopPrinter();
        rtx_compat::Device::release();
    }
    static size_t GetWidth() { return state().width; }
    static size_t GetHeight() { return state().height; }
    static size_t GetMaxSize() { return state().maxSize; }
    static bool ChangeSize(const size_t x, const size_t y)
    {
        state().width = x;
        state().height = y;
        return true;
    }
    static void SetDataInBackBuffer(const char* data, const size_t size)
    {
        State& s = state();
        {
            std::lock_guard<std::mutex> lock(s.mutex);
            std::memcpy(s.backBuffer.data(), data, size);
            s.shouldSwap = true; // FlagForBufferSwap
            s.backBufferPrintSize = size;
            s.framesSet++;
        }
        s.wake.notify_all();
    }
    static const char* GetBackBuffer() { return state().backBuffer.data(); }
    static size_t GetPrintSize() { return state().backBufferPrintSize; }
    static void UpdateRenderingFPS(const int fps) { state().renderingFps = fps; }

    // ---- the printer thread (PrintMachine::Start spawns and detaches it, PrintMachine.cpp:150-151; here it is joined
    // by StopPrinter / CleanUp).  status_lines: print the two FPS lines after every frame as the reference does.
    static void StartPrinter(int fd, bool status_lines = true)
    {
        State& s = state();
        if (s.printer.joinable()) return;
        s.fd = fd;
        s.statusLines = status_lines;
        s.terminate = false;
        s.printer = std::thread(&PrintMachine::Print);
    }
    static void StopPrinter()
    {
        State& s = state();
        if (!s.printer.joinable()) return;
        {
            std::lock_guard<std::mutex> lock(s.mutex);
            s.terminate = true; // m_terminateThread
        }
        s.wake.notify_all();
        s.printer.join();
    }
    // Blocks until every frame handed to SetDataInBackBuffer so far has been written (a lock-step caller: tests).
    static void WaitPrinted()
    {
        State& s = state();
        std::unique_lock<std::mutex> lock(s.mutex);
        s.done.wait(lock, [&] { return s.framesPrinted == s.framesSet || !s.printer.joinable() || s.terminate; });
    }
    static unsigned long long FramesPrinted() { return state().framesPrinted; }

private:
    static void write_all(int fd, const char* p, size_t n)
    {
        while (n > 0) {
            const ssize_t w = ::write(fd, p, n);
            if (w < 0) {
                if (errno == EINTR || errno == EAGAIN) continue;
                return; // the terminal went away
            }
            p += w;
            n -= (size_t)w;
        }
    }
    static bool Print() // PrintMachine.cpp:257-306
    {
        State& s = state();
        auto second = std::chrono::steady_clock::now();
        int printed_this_second = 0, printing_fps = 0;
        for (;;) {
            size_t size = 0;
            unsigned long long upto = 0;
            {
                std::unique_lock<std::mutex> lock(s.mutex);
                s.wake.wait(lock, [&] { return s.shouldSwap || s.terminate; });
                if (s.terminate && !s.shouldSwap) break;
                s.shouldSwap = false;
                size = s.backBufferPrintSize;     // m_printSize = m_backBufferPrintSize
                s.printBuffer.swap(s.backBuffer); // m_printBuffer.swap(m_backBuffer)
                upto = s.framesSet;
            }
            write_all(s.fd, "\x1b[H", 3);         // ResetConsolePointer: cursor home
            write_all(s.fd, s.printBuffer.data(), size);
            printed_this_second++;
            const auto now = std::chrono::steady_clock::now();
            if (now - second >= std::chrono::seconds(1)) {
                printing_fps = printed_this_second;
                printed_this_second = 0;
                second = now;
            }
            if (s.statusLines) {
                char line[128];
                const int n = std::snprintf(line, sizeof line, "\x1b[mRendering FPS: %d    \nPrinting FPS: %d    \n", s.renderingFps, printing_fps);
                write_all(s.fd, line, (size_t)n);
            } else {
                write_all(s.fd, "\x1b[m", 3);
            }
            {
                std::lock_guard<std::mutex> lock(s.mutex);
                s.framesPrinted = upto;
            }
            s.done.notify_all();
        }
        s.done.notify_all();
        return true;
    }
    struct State {
        size_t width = 0, height = 0, maxSize = 0, backBufferPrintSize = 0;
        std::vector<char> backBuffer, printBuffer;
        std::mutex mutex; // m_backBufferMutex
        std::condition_variable wake, done;
        bool shouldSwap = false, terminate = false, statusLines = true;
        unsigned long long framesSet = 0, framesPrinted = 0;
        int fd = 1, renderingFps = 0;
        std::thread printer;
    };
    static State& state()
    {
        static State s;
        return s;
    }
};

// ---------------------------------------------------------------------------------------------
class Scene3D {
public:
    void Init()
    {
        ctx();
        check(rtx_scene_clear(ctx()), "rtx_scene_clear");
        // the reference's start scene, Scene3D.cpp:28-33
        CreateSphere(7.0f, MyMath::Vector3(0.0f, 10.0f, 20.0f), MyMath::Vector3(255.0f, 1.0f, 1.0f));
        CreateSphere(6.0f, MyMath::Vector3(5.0f, 10.0f, 20.0f), MyMath::Vector3(1.0f, 255.0f, 1.0f));
        CreateSphere(10.0f, MyMath::Vector3(10.0f, 10.0f, 40.0f), MyMath::Vector3(1.0f, 1.0f, 255.0f));
        CreateSphere(3.0f, MyMath::Vector3(5.0f, 10.0f, 20.0f), MyMath::Vector3(225.0f, 210.0f, 20.0f));
        CreateSphere(4.0f, MyMath::Vector3(-5.0f, 10.0f, 40.0f), MyMath::Vector3(225.0f, 10.0f, 220.0f));
        CreatePlane(MyMath::Vector3(0.0f, -3.0f, 30.0f), MyMath::Vector3(0.0f, 1.0f, 0.0f), MyMath::Vector3(100.0f, 100.0f, 100.0f), 10, 20);
    }
    void Update(const long double) {} // the GPU updates objects (Scene3D.cpp:89-92)
    void CleanUp() { check(rtx_scene_clear(ctx()), "rtx_scene_clear"); }

    void CreatePlane(const MyMath::Vector3& middlePos, const MyMath::Vector3& normal, const MyMath::Vector3& color,
                     const float width, const float height)
    {
        const float p[3] = {middlePos.x, middlePos.y, middlePos.z}, n[3] = {normal.x, normal.y, normal.z}, c[3] = {color.x, color.y, color.z};
        const int idx = rtx_scene_add_plane(ctx(), p, n, c, width, height);
        if (idx < 0) check(-idx, "rtx_scene_add_plane");
    }
    void CreateSphere(const float radius, const MyMath::Vector3& middlePos, const MyMath::Vector3& color)
    {
        const float p[3] = {middlePos.x, middlePos.y, middlePos.z}, c[3] = {color.x, color.y, color.z};
        const int idx = rtx_scene_add_sphere(ctx(), p, radius, c);
        if (idx < 0) check(-idx, "rtx_scene_add_sphere");
    }
    DeviceObjectArray<Object3D*> GetObjects()
    {
        DeviceObjectArray<Object3D*> a;
        a.m_deviceArray = reinterpret_cast<Object3D**>(ctx());
        a.count = rtx_scene_count(ctx());
        a.allocatedBytes = a.count * (unsigned)sizeof(void*);
        return a;
    }

private:
    static rtx_ctx* ctx() { return rtx_compat::Device::get(PrintMachine::GetWidth(), PrintMachine::GetHeight()); }
    static void check(int status, const char* what) { rtx_compat::check(ctx(), status, what); }
};

// ---------------------------------------------------------------------------------------------
// Camera3D: the pure host math that produces the path's input (Camera3D.cpp:8-98, 207-376), through
// rtx_camera_params so that there is one implementation of it.
class Camera3D {
public:
    struct PressedKeys { // Camera3D.h:37-46
        int W = 0, A = 0, S = 0, D = 0, Shift = 0, Space = 0;
    };
    PressedKeys m_Keys;

    void Init() { refresh(); }
    void Update() { refresh(); }
    // Camera3D::Move, Camera3D.cpp:142-163: WASD in the x-z plane along the "static" right / forward vectors of the last
    // Update (:61-71; their y components take no part), Space / Shift along y; speed 10 units per second; the summed
    // direction is normalised with the safe host normalise (MyMath.h:117-123), so diagonals are not faster.
    void Move(const long double dt)
    {
        const float deltaSpeed = static_cast<float>(dt) * 10.0f;
        const MyMath::Vector3 moveX = m_staticRight * static_cast<float>(m_Keys.D - m_Keys.A);
        const MyMath::Vector3 moveZ = m_staticForward * static_cast<float>(m_Keys.W - m_Keys.S);
        MyMath::Vector3 total = moveX + moveZ;
        const float length = std::sqrt(total.x * total.x + total.y * total.y + total.z * total.z);
        const float divider = length < 0.000001f ? 0.0f : 1.0f / length;
        total = total * divider;
        m_pos.x = m_pos.x + (total.x * deltaSpeed);
        m_pos.z = m_pos.z + (total.z * deltaSpeed);
        m_pos.y += (float)(m_Keys.Space - m_Keys.Shift) * deltaSpeed;
    }
    // Camera3D::AddRot, Camera3D.cpp:166-187: p / y / r are mouse counts; 0.002 rad per count whatever dt; pitch clamped
    // just inside +-pi/2.
    void AddRot(const long double, const short p, const short y, const short r)
    {
        const float deltaSpeed = 0.002f;
        m_rot.x -= ((float)p * deltaSpeed);
        m_rot.y += ((float)y * deltaSpeed);
        m_rot.z += ((float)r * deltaSpeed);
        if (m_rot.x > static_cast<float>(3.14159265358979323846 / 2.0)) m_rot.x = static_cast<float>((3.14159265358979323846 / 2.0) - 0.0001);
        if (m_rot.x < static_cast<float>(-3.14159265358979323846 / 2.0)) m_rot.x = static_cast<float>((-3.14159265358979323846 / 2.0) + 0.0001);
    }
    void SetRot(const float p, const float y, const float r)
    {
        m_rot = MyMath::Vector3(p, y, r);
    }
    void SetPos(const float x, const float y, const float z) { m_pos = MyMath::Vector3(x, y, z); }
    const MyMath::Matrix GetInverseVMatrix() const { return m_inverse; }
    const MyMath::Matrix& GetPMatrix() const { return m_pMatrix; }
    const MyMath::Vector3& GetPos() const { return m_pos; }
    const MyMath::Vector3& GetRot() const { return m_rot; }
    float GetFarPlaneDistance() const { return m_far; }

private:
    void refresh()
    {
        rtx_params p;
        const float pos[3] = {m_pos.x, m_pos.y, m_pos.z}, rot[3] = {m_rot.x, m_rot.y, m_rot.z};
        if (rtx_camera_params(PrintMachine::GetWidth(), PrintMachine::GetHeight(), pos, rot, &p) != RTX_OK) {
            throw std::runtime_error("rtx_camera_params: singular view matrix or PrintMachine not started");
        }
        m_inverse = MyMath::Matrix(MyMath::Vector4(p.inv_v[0], p.inv_v[1], p.inv_v[2], p.inv_v[3]),
                                   MyMath::Vector4(p.inv_v[4], p.inv_v[5], p.inv_v[6], p.inv_v[7]),
                                   MyMath::Vector4(p.inv_v[8], p.inv_v[9], p.inv_v[10], p.inv_v[11]),
                                   MyMath::Vector4(p.inv_v[12], p.inv_v[13], p.inv_v[14], p.inv_v[15]));
        m_pMatrix.row1.x = p.element1;
        m_pMatrix.row2.y = p.element2;
        m_far = p.cam_far;
        // Camera3D.cpp:61-71
        const float yaw = m_rot.y;
        m_staticForward = MyMath::Vector3(-std::sin(yaw), -std::cos(yaw), -std::cos(yaw));
        m_staticRight = MyMath::Vector3(std::cos(yaw), -std::sin(yaw), -std::sin(yaw));
    }
    MyMath::Matrix m_inverse, m_pMatrix;
    MyMath::Vector3 m_staticForward, m_staticRight;
    MyMath::Vector3 m_pos;
    MyMath::Vector3 m_rot = MyMath::Vector3(0.0f, 3.14159274101257324f, 0.0f); // Camera3D.h:62
    float m_far = 250.0f;
};

// ---------------------------------------------------------------------------------------------
class RayTracing {
public:
    RayTracing() = delete;
    // params: HOST pointer here (device pointer in the reference); resultArray: device memory, pre-zeroed by
    // the caller as in the reference; gridDims/blockDims are accepted and ignored.
    static void RayTrace(const dim3_compat&, const dim3_compat&, Object3D** const objects, const unsigned int count,
                         const RayTracingCPUToGPUData* params, char* resultArray, const RenderingMode mode)
    {
        rtx_ctx* ctx = reinterpret_cast<rtx_ctx*>(objects);
        if (!ctx || !params) throw std::runtime_error("RayTracing::RayTrace: null scene handle or params");
        if (count != rtx_scene_count(ctx)) throw std::runtime_error("RayTracing::RayTrace: count does not match the scene");
        const rtx_params p = rtx_compat::to_rtx_params(*params);
        rtx_compat::check(ctx, rtx_render_rows(ctx, &p, (int)mode, 0, (size_t)p.y, resultArray, 0, nullptr, RTX_RENDER_DEFAULT),
                          "rtx_render_rows");
    }
};

// ---------------------------------------------------------------------------------------------
class RayTracingManager {
public:
    RayTracingManager()
    {
        // the constructor sizes its buffers from PrintMachine::GetMaxSize() (RayTracingManager.cu:58-66)
        if (PrintMachine::GetMaxSize() == 0) throw std::runtime_error("PrintMachine::Start must precede RayTracingManager()");
        m_ctx = rtx_compat::Device::get(PrintMachine::GetWidth(), PrintMachine::GetHeight());
        // pinned, so that the copy of the minimised stream runs at PCIe rate (the reference's is pageable)
        m_minimizedResultArray = static_cast<char*>(rtx_host_alloc(m_ctx, PrintMachine::GetMaxSize()));
        if (!m_minimizedResultArray) throw std::runtime_error(std::string("rtx_host_alloc: ") + rtx_last_error(m_ctx));
    }
    ~RayTracingManager()
    {
        // the shared context itself is released by PrintMachine::CleanUp()
        if (rtx_compat::Device::slot() == m_ctx) rtx_host_free(m_ctx, m_minimizedResultArray);
    }

    // Synchronous, like the reference: on return the minimised frame is in PrintMachine's back buffer.
    void Update(const RayTracingCPUToGPUData& params, const DeviceObjectArray<Object3D*>& deviceObjects, double dt)
    {
        if (reinterpret_cast<rtx_ctx*>(deviceObjects.m_deviceArray) != m_ctx) {
            throw std::runtime_error("RayTracingManager::Update: objects do not belong to this device context");
        }
        const rtx_params p = rtx_compat::to_rtx_params(params);
        size_t newSize = 0;
        rtx_compat::check(m_ctx, rtx_update(m_ctx, &p, (int)currentRenderingMode, dt, /*run_physics=*/1,
                                            m_minimizedResultArray, &newSize), "rtx_update");
        PrintMachine::SetDataInBackBuffer(m_minimizedResultArray, newSize);
    }
    void SetRenderingMode(const RenderingMode newRenderMode) { currentRenderingMode = newRenderMode; }

private:
    rtx_ctx* m_ctx = nullptr;
    char* m_minimizedResultArray = nullptr;
    RenderingMode currentRenderingMode = BIT_ASCII; // RayTracingManager.h:53
};
