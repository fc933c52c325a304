// headless_engine.cpp -- what Engine3D::Start + Engine3D::Render do (Engine3D.cpp:9-28, 81-107),
// written against include/rtx_compat.hpp: the reference's class names and call order, no console.
//
//   headless_engine <W> <H> <frames> <mode 0..4> <dt> <out_file>
// Renders `frames` frames of the reference's start scene and writes the last minimised frame (what
// the reference's printer thread would fwrite to stdout, PrintMachine.cpp:289-290) to out_file.
#include "rtx_compat.hpp"

#include <cstdio>
#include <cstdlib>
#include <memory>

int main(int argc, char** argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s W H frames mode dt out_file\n", argv[0]);
        return 2;
    }
    const size_t W = std::strtoul(argv[1], nullptr, 10), H = std::strtoul(argv[2], nullptr, 10);
    const int frames = std::atoi(argv[3]);
    const RenderingMode mode = (RenderingMode)std::atoi(argv[4]);
    const double dt = std::atof(argv[5]);
    try {
        // Engine3D::Start
        PrintMachine::Start(W, H);
        auto rayTracingManager = std::make_unique<RayTracingManager>();
        auto camera = std::make_unique<Camera3D>();
        auto scene = std::make_unique<Scene3D>();
        camera->Init();
        camera->Update();
        scene->Init();
        rayTracingManager->SetRenderingMode(mode);

        for (int f = 0; f < frames; f++) {
            // Engine3D::Render
            camera->Update();
            scene->Update(dt);
            RayTracingCPUToGPUData params;
            params.inverseVMatrix = camera->GetInverseVMatrix();
            params.camPos = camera->GetPos();
            params.x = PrintMachine::GetWidth();
            params.y = PrintMachine::GetHeight();
            params.element1 = camera->GetPMatrix().row1.x;
            params.element2 = camera->GetPMatrix().row2.y;
            params.camFarDist = camera->GetFarPlaneDistance();
            DeviceObjectArray<Object3D*> objects = scene->GetObjects();
            rayTracingManager->Update(params, objects, dt);
        }

        std::FILE* f = std::fopen(argv[6], "wb");
        if (!f) {
            std::perror("fopen");
            return 1;
        }
        std::fwrite(PrintMachine::GetBackBuffer(), 1, PrintMachine::GetPrintSize(), f);
        std::fclose(f);
        std::printf("frames=%d bytes=%zu\n", frames, PrintMachine::GetPrintSize());
        rayTracingManager.reset();
        PrintMachine::CleanUp();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
