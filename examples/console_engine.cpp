// console_engine.cpp -- the reference's Engine3D loop on a POSIX terminal, written against
// include/rtx_compat.hpp: Engine3D::Start / Run / CheckKeyboard / Render (Engine3D.cpp:9-107, 110-240) with termios
// raw-mode keyboard input where the reference polls Win32 key state, and PrintMachine's printer thread
// (PrintMachine.cpp:257-306) writing each minimised frame to the terminal after an ANSI cursor-home.
//
//   console_engine [W H] [--mode 0..4] [--frames N] [--dt seconds] [--lockstep] [--no-spawn] [--no-status] [--trace FILE] [--keys-only]
//
// Keys (the reference's, as far as a terminal can deliver them -- there are no key-up events and no mouse, so a key
// counts as held for the frame in which its byte arrives and the arrow keys stand in for mouse motion):
//   w a s d        Camera3D::m_Keys.W/A/S/D for this frame          (Engine3D.cpp:113-147)
//   space, z       m_Keys.Space (up), m_Keys.Shift (down)           (:155-171; a terminal cannot see Shift alone)
//   arrow keys     Camera3D::AddRot by 25 mouse counts = 0.05 rad   (:209-239)
//   1..5, F1..F5   RayTracingManager::SetRenderingMode(BIT_ASCII .. RGB_NORMALS)   (:178-197)
//   Esc, x         quit                                             (:173-176)
// Once per second of engine time a random sphere is created, as Engine3D::Run does (Engine3D.cpp:60-69; rand() is
// never seeded there either).  --lockstep: exactly one key event is consumed per frame (blocking; end of input
// quits) and the loop waits until the printer has written the frame; with --dt this makes a run reproducible, and
// --trace FILE records what each frame was rendered with (hex floats) so that a checker can re-render it.
#include "rtx_compat.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include <poll.h>
#include <termios.h>
#include <unistd.h>

namespace {

struct Terminal { // raw input mode for the lifetime of the object; output processing (NL -> CR NL) stays on
    termios saved{};
    bool active = false;
    Terminal()
    {
        if (isatty(STDIN_FILENO) && tcgetattr(STDIN_FILENO, &saved) == 0) {
            termios raw = saved;
            raw.c_lflag &= ~(tcflag_t)(ICANON | ECHO | ISIG);
            raw.c_iflag &= ~(tcflag_t)(IXON | ICRNL);
            raw.c_cc[VMIN] = 0;
            raw.c_cc[VTIME] = 0;
            active = tcsetattr(STDIN_FILENO, TCSANOW, &raw) == 0;
        }
    }
    ~Terminal()
    {
        if (active) tcsetattr(STDIN_FILENO, TCSANOW, &saved);
    }
};

enum Key { K_NONE, K_W, K_A, K_S, K_D, K_SPACE, K_SHIFT, K_UP, K_DOWN, K_LEFT, K_RIGHT, K_MODE0, K_MODE1, K_MODE2, K_MODE3, K_MODE4, K_QUIT, K_EOF, K_OTHER };

// Bytes from stdin -> key events.  Escape sequences: CSI A/B/C/D arrows; SS3 P/Q/R/S and CSI 11~..15~ for F1..F5.
struct KeyReader {
    std::string pending;
    bool eof = false;
    void fill(int timeout_ms)
    {
        pollfd pf{STDIN_FILENO, POLLIN, 0};
        if (poll(&pf, 1, timeout_ms) > 0) {
            char buf[256];
            const ssize_t n = read(STDIN_FILENO, buf, sizeof buf);
            if (n > 0) pending.append(buf, (size_t)n);
            else if (n == 0 || (pf.revents & (POLLHUP | POLLERR))) eof = true;
        }
    }
    Key next()
    {
        if (pending.empty()) return eof ? K_EOF : K_NONE;
        const unsigned char c = (unsigned char)pending[0];
        if (c == 0x1b) {
            if (pending.size() == 1) {
                fill(20); // a lone ESC is the Escape key; a sequence arrives in one burst
                if (pending.size() == 1) {
                    pending.erase(0, 1);
                    return K_QUIT;
                }
            }
            if (pending[1] == '[' || pending[1] == 'O') {
                size_t i = 2;
                while (i < pending.size() && ((pending[i] >= '0' && pending[i] <= '9') || pending[i] == ';')) i++;
                if (i >= pending.size()) {
                    pending.clear();
                    return K_OTHER;
                }
                const std::string arg = pending.substr(2, i - 2);
                const char fin = pending[i];
                pending.erase(0, i + 1);
                switch (fin) {
                case 'A': return K_UP;
                case 'B': return K_DOWN;
                case 'C': return K_RIGHT;
                case 'D': return K_LEFT;
                case 'P': return K_MODE0;
                case 'Q': return K_MODE1;
                case 'R': return K_MODE2;
                case 'S': return K_MODE3;
                case '~':
                    if (arg == "11") return K_MODE0;
                    if (arg == "12") return K_MODE1;
                    if (arg == "13") return K_MODE2;
                    if (arg == "14") return K_MODE3;
                    if (arg == "15") return K_MODE4;
                    return K_OTHER;
                default: return K_OTHER;
                }
            }
            pending.erase(0, 1);
            return K_QUIT;
        }
        pending.erase(0, 1);
        switch (c) {
        case 'w': case 'W': return K_W;
        case 'a': case 'A': return K_A;
        case 's': case 'S': return K_S;
        case 'd': case 'D': return K_D;
        case ' ': return K_SPACE;
        case 'z': case 'Z': return K_SHIFT;
        case '1': return K_MODE0;
        case '2': return K_MODE1;
        case '3': return K_MODE2;
        case '4': return K_MODE3;
        case '5': return K_MODE4;
        case 'x': case 'X': case 3 /* ^C in raw mode */: return K_QUIT;
        default: return K_OTHER;
        }
    }
};

} // namespace

int main(int argc, char** argv)
{
    size_t W = 160, H = 50;
    int mode0 = BIT_ASCII, max_frames = -1; // the reference starts in BIT_ASCII (RayTracingManager.h:53)
    double fixed_dt = -1.0;
    bool lockstep = false, spawn = true, status = true, keys_only = false;
    const char* trace_path = nullptr;
    int positional = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--mode" && i + 1 < argc) mode0 = std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) max_frames = std::atoi(argv[++i]);
        else if (a == "--dt" && i + 1 < argc) fixed_dt = std::atof(argv[++i]);
        else if (a == "--lockstep") lockstep = true;
        else if (a == "--no-spawn") spawn = false;
        else if (a == "--no-status") status = false;
        else if (a == "--keys-only") keys_only = true; // input check without a GPU: raw mode on, print each decoded key, quit on x / Esc / end of input
        else if (a == "--trace" && i + 1 < argc) trace_path = argv[++i];
        else if (a[0] != '-' && positional == 0) { W = std::strtoul(argv[i], nullptr, 10); positional++; }
        else if (a[0] != '-' && positional == 1) { H = std::strtoul(argv[i], nullptr, 10); positional++; }
        else {
            std::fprintf(stderr, "usage: %s [W H] [--mode 0..4] [--frames N] [--dt s] [--lockstep] [--no-spawn] [--no-status] [--trace FILE]\n", argv[0]);
            return 2;
        }
    }
    if (W == 0 || H == 0 || mode0 < 0 || mode0 > 4) {
        std::fprintf(stderr, "bad size or mode\n");
        return 2;
    }
    if (keys_only) {
        static const char* const names[] = {"none", "w", "a", "s", "d", "space", "shift", "up", "down", "left", "right", "mode0", "mode1", "mode2", "mode3",
                                            "mode4", "quit", "eof", "other"};
        Terminal term;
        KeyReader keys;
        std::printf("raw %d\n", term.active ? 1 : 0);
        std::fflush(stdout);
        for (;;) {
            Key k = keys.next();
            while (k == K_NONE) {
                keys.fill(1000);
                k = keys.next();
            }
            std::printf("key %s\n", names[k]);
            std::fflush(stdout);
            if (k == K_QUIT || k == K_EOF) break;
        }
        return 0;
    }
    std::FILE* trace = trace_path ? std::fopen(trace_path, "w") : nullptr;
    try {
        Terminal term;
        // Engine3D::Start (Engine3D.cpp:9-28): printer first (the manager sizes its buffers from it), then the rest
        PrintMachine::Start(W, H);
        auto rayTracingManager = std::make_unique<RayTracingManager>();
        auto camera = std::make_unique<Camera3D>();
        auto scene = std::make_unique<Scene3D>();
        camera->Init();
        camera->Update();
        scene->Init();
        rayTracingManager->SetRenderingMode((RenderingMode)mode0);
        int mode = mode0;
        const char hide[] = "\x1b[?25l\x1b[2J"; // hide the cursor, clear (PrintMachine.cpp:120)
        if (write(STDOUT_FILENO, hide, sizeof hide - 1) < 0) return 1;
        PrintMachine::StartPrinter(STDOUT_FILENO, status);

        KeyReader keys;
        auto last = std::chrono::steady_clock::now();
        double fpsTimer = 0.0;
        int fps = 0, frame = 0;
        bool quit = false;
        while (!quit && (max_frames < 0 || frame < max_frames)) { // Engine3D::Run
            const auto now = std::chrono::steady_clock::now();
            const double dt = fixed_dt >= 0.0 ? fixed_dt : std::chrono::duration<double>(now - last).count();
            last = now;
            fpsTimer += dt;
            fps++;

            // CheckKeyboard: key state for this frame
            camera->m_Keys = Camera3D::PressedKeys();
            short rotP = 0, rotY = 0;
            auto apply = [&](Key k) {
                switch (k) {
                case K_W: camera->m_Keys.W = 1; break;
                case K_A: camera->m_Keys.A = 1; break;
                case K_S: camera->m_Keys.S = 1; break;
                case K_D: camera->m_Keys.D = 1; break;
                case K_SPACE: camera->m_Keys.Space = 1; break;
                case K_SHIFT: camera->m_Keys.Shift = 1; break;
                case K_UP: rotP = (short)(rotP + 25); break;
                case K_DOWN: rotP = (short)(rotP - 25); break;
                case K_LEFT: rotY = (short)(rotY + 25); break;
                case K_RIGHT: rotY = (short)(rotY - 25); break;
                case K_MODE0: case K_MODE1: case K_MODE2: case K_MODE3: case K_MODE4:
                    mode = (int)k - (int)K_MODE0;
                    rayTracingManager->SetRenderingMode((RenderingMode)mode);
                    break;
                case K_QUIT: case K_EOF: quit = true; break;
                default: break;
                }
            };
            if (lockstep) {
                Key k = keys.next();
                while (k == K_NONE) {
                    keys.fill(1000);
                    k = keys.next();
                }
                apply(k);
            } else {
                keys.fill(0);
                for (Key k = keys.next(); k != K_NONE; k = keys.next()) {
                    apply(k);
                    if (k == K_EOF) break;
                }
            }
            if (quit) break;
            camera->AddRot(dt, rotP, rotY, 0);
            camera->Move(dt); // Engine3D.cpp:56

            // Engine3D::Render (Engine3D.cpp:81-107)
            camera->Update();
            RayTracingCPUToGPUData params;
            params.inverseVMatrix = camera->GetInverseVMatrix();
            params.camPos = camera->GetPos();
            params.x = PrintMachine::GetWidth();
            params.y = PrintMachine::GetHeight();
            params.element1 = camera->GetPMatrix().row1.x;
            params.element2 = camera->GetPMatrix().row2.y;
            params.camFarDist = camera->GetFarPlaneDistance();
            rayTracingManager->Update(params, scene->GetObjects(), dt);
            if (trace) {
                const MyMath::Vector3 p = camera->GetPos(), r = camera->GetRot();
                std::fprintf(trace, "frame %d mode %d dt %a pos %a %a %a rot %a %a %a bytes %zu\n", frame, mode, dt, (double)p.x, (double)p.y, (double)p.z,
                             (double)r.x, (double)r.y, (double)r.z, PrintMachine::GetPrintSize());
                std::fflush(trace);
            }
            if (lockstep) PrintMachine::WaitPrinted();

            if (fpsTimer >= 1.0) { // Engine3D.cpp:60-69
                if (spawn) {
                    const float radius = static_cast<float>(rand() % 10);
                    const MyMath::Vector3 pos(rand() % 100 - 50, rand() % 100 - 50, rand() % 100 - 50);
                    const MyMath::Vector3 col(rand() % 255, rand() % 255, rand() % 255);
                    scene->CreateSphere(radius, pos, col);
                    if (trace) {
                        std::fprintf(trace, "spawn %a %a %a %a %a %a %a\n", (double)radius, (double)pos.x, (double)pos.y, (double)pos.z, (double)col.x,
                                     (double)col.y, (double)col.z);
                    }
                }
                PrintMachine::UpdateRenderingFPS(fps);
                fpsTimer = 0.0;
                fps = 0;
            }
            frame++;
        }
        PrintMachine::StopPrinter();
        const char show[] = "\x1b[m\x1b[?25h\n"; // reset colour, show the cursor (PrintMachine.cpp:154-166)
        if (write(STDOUT_FILENO, show, sizeof show - 1) < 0) return 1;
        rayTracingManager.reset();
        PrintMachine::CleanUp();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        if (trace) std::fclose(trace);
        return 1;
    }
    if (trace) std::fclose(trace);
    return 0;
}
