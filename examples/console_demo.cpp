// console_demo.cpp -- the reference's main loop on a POSIX terminal: what
// Engine3D::Run (Engine3D.cpp:30-79) and the printer thread (PrintMachine.cpp:257-306) do together,
// with the camera on a fixed orbit instead of Win32 keyboard/mouse input.  Each frame goes through
// RayTracingManager::Update (include/rtx_compat.hpp -> librtx_hip.so) and the minimised ANSI stream is
// written to stdout after a cursor-home escape (the reference calls SetConsoleCursorPosition(0,0)).
//
//   console_demo [W H frames mode]      e.g.  console_demo 160 50 300 2     (mode: 0..4, RayTracingManager.h:21)
#include "rtx_compat.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

int main(int argc, char** argv)
{
    const size_t W = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 160;
    const size_t H = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 50;
    const int frames = argc > 3 ? std::atoi(argv[3]) : 200;
    const RenderingMode mode = (RenderingMode)(argc > 4 ? std::atoi(argv[4]) : 2);
    try {
        PrintMachine::Start(W, H);
        auto manager = std::make_unique<RayTracingManager>();
        auto camera = std::make_unique<Camera3D>();
        auto scene = std::make_unique<Scene3D>();
        camera->Init();
        scene->Init();
        manager->SetRenderingMode(mode);
        std::printf("\x1b[?25l\x1b[2J"); // hide the cursor, clear (PrintMachine.cpp:120)

        auto last = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
            const auto now = std::chrono::steady_clock::now();
            const double dt = std::chrono::duration<double>(now - last).count();
            last = now;
            // a slow sway of the camera in front of the start scene
            const float phase = 0.02f * (float)f;
            camera->SetPos(6.0f * std::sin(phase), 4.0f + 2.0f * std::sin(0.5f * phase), -4.0f + 3.0f * std::cos(phase));
            camera->SetRot(0.0f, 3.14159274f - 0.15f * std::sin(phase), 0.0f);
            camera->Update();

            RayTracingCPUToGPUData params;
            params.inverseVMatrix = camera->GetInverseVMatrix();
            params.camPos = camera->GetPos();
            params.x = PrintMachine::GetWidth();
            params.y = PrintMachine::GetHeight();
            params.element1 = camera->GetPMatrix().row1.x;
            params.element2 = camera->GetPMatrix().row2.y;
            params.camFarDist = camera->GetFarPlaneDistance();
            manager->Update(params, scene->GetObjects(), dt);

            std::fputs("\x1b[H", stdout); // cursor home
            std::fwrite(PrintMachine::GetBackBuffer(), 1, PrintMachine::GetPrintSize(), stdout);
            std::fflush(stdout);
        }
        std::printf("\x1b[m\x1b[?25h\n"); // reset colour, show the cursor (PrintMachine.cpp:154-166)
        manager.reset();
        PrintMachine::CleanUp();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
