// Unit tests of csrc/rtx_plan.hpp -- the pure planning behind rtx_render_rows -- on the CPU:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/host/test_plan.cpp -o test_plan && ./test_plan
// (tests/test_host_plan.py builds and runs it).  No HIP, no GPU.
#include "../../raytracing-in-windows-console_amd/csrc/rtx_plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace rtxplan;

static int g_failed = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
            g_failed++;                                                          \
        }                                                                        \
    } while (0)

// the reference camera at 1080p: yaw pi, element1 = 0.57735 * 1080 / 100, element2 = 0.57735
static void yaw_matrix(double yaw, float m[16])
{
    const float c = (float)std::cos(yaw), s = (float)std::sin(yaw);
    const float r[16] = {c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1};
    std::memcpy(m, r, sizeof r);
}

static TileRequest request(uint64_t W, uint64_t H, uint64_t rows, uint32_t ns, bool cull)
{
    float m[16];
    yaw_matrix(3.14159265, m);
    double sx, sy;
    pixel_steps(m, 0.57735f * (float)H / 100.0f, 0.57735f, W, H, &sx, &sy);
    TileRequest q;
    q.W = W;
    q.H = H;
    q.rows = rows;
    q.ns = ns;
    q.aspect = sx / sy;
    q.n_cu = 256;
    q.cull = cull;
    return q;
}

static void test_tile_shapes()
{
    // config 2: 8100 tiles of 256 pixels over 1792 resident slots -> 5 sub-tiles, one dispatch round
    TileShape t = plan_tiles(request(1920, 1080, 1080, 1025, true));
    CHECK(t.nsub == 5 && !t.refine);
    CHECK((uint64_t)t.grid_x * t.grid_y <= resident_slots(256));
    CHECK(t.mw <= 128 && t.mh <= 128 && t.mw * t.mh == 256u * t.nsub);
    CHECK((uint64_t)t.grid_x * t.mw >= 1920 && (uint64_t)t.grid_y * t.mh >= 1080);
    // config 5: dense -> 2 sub-tiles, per-wave refinement, macro tile within 64 x 64
    t = plan_tiles(request(1920, 1080, 1080, 65536, true));
    CHECK(t.nsub == 2 && t.refine && t.mw <= 64 && t.mh <= 64);
    {
        // ... with frames in flight on several streams: 4 sub-tiles (the set-up shared by twice the pixels), still refined, still
        // within the REFINE kernels' 64 x 64 tables; a view that is only locally dense keeps 2
        TileRequest q = request(1920, 1080, 1080, 65536, true);
        q.in_flight = true;
        const TileShape f = plan_tiles(q);
        CHECK(f.nsub == 4 && f.refine && f.mw <= 64 && f.mh <= 64 && f.mw * f.mh == 1024);
        TileRequest v = request(1920, 1080, 1080, 1025, true);
        v.in_flight = true;
        v.view_dense = true;
        CHECK(plan_tiles(v).nsub == 2 && plan_tiles(v).refine);
        v.view_dense = false;
        CHECK(plan_tiles(v).nsub == 5);
    }
    // config 3 (4K): 8 sub-tiles; sub-tiles at least 8 pixels wide whatever the pixel aspect
    t = plan_tiles(request(3840, 2160, 2160, 4102, true));
    CHECK(t.nsub == 8 && t.lw >= 3 && !t.refine);
    // config 4 (8K): 8 sub-tiles, several rounds
    t = plan_tiles(request(7680, 4320, 4320, 1024, true));
    CHECK(t.nsub == 8 && t.lw >= 3 && (uint64_t)t.grid_x * t.grid_y > resident_slots(256));
    // the slabs of a sharded 1080p frame: the smallest count that keeps one round
    CHECK(plan_tiles(request(1920, 1080, 540, 1025, true)).nsub == 3);
    CHECK(plan_tiles(request(1920, 1080, 270, 1025, true)).nsub == 2);
    CHECK(plan_tiles(request(1920, 1080, 135, 1025, true)).nsub == 1);
    // brute: 64 x 4 sub-tiles
    t = plan_tiles(request(400, 150, 150, 6, false));
    CHECK(t.lw == 6 && t.nsub == 1 && t.mw == 64 && t.mh == 4);
    // every shape the options can ask for stays within the kernel's limits
    for (int sub : {1, 2, 3, 4, 5, 6, 8, 16}) {
        for (int lw = 2; lw <= 6; lw++) {
            TileRequest q = request(1000, 700, 700, 5000, true);
            q.opt_subtiles = sub;
            q.opt_tile_log2w = lw;
            t = plan_tiles(q);
            CHECK(t.nsub == (uint32_t)sub && t.mw * t.mh == 256u * t.nsub);
            if ((sub & (sub - 1)) == 0) CHECK(t.mw <= 128 && t.mh <= 128);
        }
    }
    // degenerate frames
    t = plan_tiles(request(1, 1, 1, 100, true));
    CHECK(t.grid_x == 1 && t.grid_y == 1);
}

static void test_cell_grid()
{
    const TileRequest q = request(1920, 1080, 1080, 65536, true);
    const TileShape t = plan_tiles(q);
    const CellGrid g = plan_cells(t, 65536, q.aspect, 0);
    CHECK(g.n_cells == g.cells_x * g.cells_y && g.n_cells <= 1024 && g.n_cells >= 256);
    CHECK(g.cap == 4u * 65536u / g.n_cells + 1024u);
    CHECK((uint64_t)g.cells_x << g.gx >= t.grid_x && (uint64_t)g.cells_y << g.gy >= t.grid_y);
    CHECK(g.splits >= 1 && g.splits * g.n_blocks <= 1024 + g.n_blocks);
    CHECK(g.cell_w == t.mw << g.gx && g.cell_h == t.mh << g.gy);
    // the scratch is O(spheres): 4 ns + 1024 cells words
    CHECK((uint64_t)g.n_cells * g.cap <= 4ull * 65536 + 1024ull * g.n_cells);
    // small scenes: the capacity never exceeds the scene; an explicit capacity is taken as is
    const CellGrid s = plan_cells(t, 100, q.aspect, 0);
    CHECK(s.cap == 100 && s.splits == 1);
    CHECK(plan_cells(t, 65536, q.aspect, 7).cap == 7);
}

static void test_xcd_order()
{
    for (auto dims : std::vector<std::vector<uint32_t>>{{30, 135, 0, 2}, {64, 64, 1, 1}, {7, 9, 2, 1}, {120, 34, 2, 0}, {1, 64, 0, 0}, {100, 1, 3, 0}}) {
        const uint32_t gx_n = dims[0], gy_n = dims[1], gx = dims[2], gy = dims[3], n = gx_n * gy_n;
        std::vector<uint32_t> order(n, 0xffffffffu);
        xcd_cell_order(gx_n, gy_n, gx, gy, order.data());
        // a permutation of the tiles
        std::vector<int> seen(n, 0);
        for (uint32_t b = 0; b < n; b++) {
            const uint32_t bx = order[b] & 0xffffu, by = order[b] >> 16;
            CHECK(bx < gx_n && by < gy_n);
            if (bx < gx_n && by < gy_n) seen[by * gx_n + bx]++;
        }
        CHECK(std::all_of(seen.begin(), seen.end(), [](int c) { return c == 1; }));
        // the blocks that share an XCD (equal b % 8) hold whole cells: few cells are split over two labels, every label sees about
        // an eighth of the cells, and those come from all over the frame (every band of cell rows is represented)
        const uint32_t cells_x = (gx_n + (1u << gx) - 1) >> gx, cells_y = (gy_n + (1u << gy) - 1) >> gy;
        const uint32_t n_cells = cells_x * cells_y;
        if (n < 64) continue;
        std::vector<uint32_t> labels_of_cell(n_cells, 0);
        for (uint32_t b = 0; b < n; b++) {
            const uint32_t cell = ((order[b] >> 16) >> gy) * cells_x + ((order[b] & 0xffffu) >> gx);
            labels_of_cell[cell] |= 1u << (b % 8);
        }
        uint32_t split = 0;
        for (uint32_t m : labels_of_cell) split += (m & (m - 1)) != 0;
        CHECK(split <= n_cells / 8 + 8);
        for (uint32_t x = 0; x < 8; x++) {
            uint32_t cells = 0, rows_seen = 0;
            std::vector<int> row_has(cells_y, 0);
            for (uint32_t c = 0; c < n_cells; c++) {
                if (labels_of_cell[c] & (1u << x)) {
                    cells++;
                    row_has[c / cells_x] = 1;
                }
            }
            for (int r : row_has) rows_seen += (uint32_t)r;
            CHECK(cells <= n_cells / 8 + n_cells / 16 + 8);
            if (cells_x >= 8) CHECK(rows_seen == cells_y);
        }
    }
}

static View yaw_view(double yaw, float px = 0, float py = 0, float pz = 0)
{
    float m[16];
    yaw_matrix(yaw, m);
    const float pos[3] = {px, py, pz};
    return view_of(m, pos);
}

static void test_dispatch_order_static_view()
{
    DispatchOrder o;
    const View v = yaw_view(3.14159265);
    int sorts = 0, switches = 0, since_sort = -1000;
    bool used_before_switch = false;
    for (int f = 0; f < 400; f++) {
        const DispatchOrder::Decision d = o.next(v, 0.0, true, -1);
        if (d.switch_order) {
            switches++;
            CHECK(since_sort >= DispatchOrder::kLag - 1); // the launches between a pass and the switch to its order
        }
        if (d.use_order && switches == 0) used_before_switch = true;
        if (d.sort_now) {
            CHECK(d.leave_estimates && d.balance);
            CHECK(!o.pass_pending());                     // never two passes in flight
            CHECK(d.prev_is_order == d.use_order);        // the pass is told the order the launch really ran under
            sorts++;
            since_sort = 0;
        } else {
            CHECK(!d.leave_estimates);
            since_sort++;
        }
        if (d.use_order) CHECK(d.half == o.current_half());
        o.launched(d);
    }
    CHECK(!used_before_switch);
    CHECK(sorts >= 16 && sorts <= 40);   // one in three launches for the first 64, then every 64th
    CHECK(switches == sorts || switches == sorts - 1);
    CHECK(o.have_order());
}

static void test_dispatch_order_moving_views()
{
    // a camera that creeps: the order goes stale and is refreshed, and a stale order is never used
    {
        DispatchOrder o;
        int stale_used = 0, sorts = 0;
        double measured_at = 0.0, pending_at = 0.0;
        for (int f = 0; f < 2000; f++) {
            const double yaw = 3.14159265 + 1.0e-4 * f;
            const DispatchOrder::Decision d = o.next(yaw_view(yaw), 0.0, true, -1);
            if (d.switch_order) measured_at = pending_at;
            if (d.use_order && std::fabs(yaw - measured_at) > 1.5 * DispatchOrder::kNear) stale_used++;
            if (d.sort_now) {
                sorts++;
                pending_at = yaw;
            }
            o.launched(d);
        }
        CHECK(stale_used == 0);
        CHECK(sorts > 20); // kept up with the camera
    }
    // a camera that jumps every frame: nothing is derived, nothing is used
    {
        DispatchOrder o;
        int sorts = 0, used = 0;
        for (int f = 0; f < 300; f++) {
            const DispatchOrder::Decision d = o.next(yaw_view(3.14159265 + 0.01 * f), 0.0, true, -1);
            // (the very first launch has no previous view to compare with other than the zero view: it may sort once)
            sorts += d.sort_now && f > 0;
            used += d.use_order;
            o.launched(d);
        }
        CHECK(sorts == 0 && used == 0);
    }
    // spheres that move (scene drift) age an order like a camera that moves
    {
        DispatchOrder o;
        const View v = yaw_view(3.14159265);
        int used_late = 0;
        for (int f = 0; f < 200; f++) {
            const double drift = f < 100 ? 0.0 : 0.5 * (f - 99); // physics starts at frame 100: half a unit per frame
            const DispatchOrder::Decision d = o.next(v, drift, true, -1);
            if (f > 110 && d.use_order) used_late++;
            o.launched(d);
        }
        CHECK(used_late == 0);
    }
    // grids of several rounds: in-line sort after the 1st and 2nd launch, then every 16th; the order is used from then on
    {
        DispatchOrder o;
        std::vector<int> sorted_at;
        for (int f = 0; f < 40; f++) {
            const DispatchOrder::Decision d = o.next(yaw_view(3.14159265), 0.0, false, -1);
            CHECK(!d.balance && !d.switch_order);
            if (d.sort_now) sorted_at.push_back(f);
            CHECK(d.use_order == (f >= 1));
            o.launched(d);
        }
        CHECK((sorted_at == std::vector<int>{0, 1, 15, 31}));
    }
    // reset with a pass pending (the grid changed): starts over cleanly
    {
        DispatchOrder o;
        DispatchOrder::Decision d = o.next(yaw_view(3.14159265), 0.0, true, -1); // (no view before it to compare with: no pass yet)
        o.launched(d);
        d = o.next(yaw_view(3.14159265), 0.0, true, -1);
        CHECK(d.sort_now);
        o.launched(d);
        CHECK(o.pass_pending());
        o.reset();
        CHECK(!o.pass_pending() && !o.have_order() && o.frames() == 0);
        d = o.next(yaw_view(3.14159265), 0.0, true, -1);
        CHECK(!d.switch_order && !d.use_order && d.sort_now);
        // spectral norm of a rotation difference: 2 sin(phi / 2), not sqrt(2) times it
        const View va = yaw_view(1.0), vb = yaw_view(1.01);
        double dm[9];
        for (int k = 0; k < 9; k++) dm[k] = (double)vb.rot[k] - (double)va.rot[k];
        const double sn = spectral_norm3(dm);
        CHECK(sn >= 2.0 * std::sin(0.005) * 0.9999 && sn <= 2.0 * std::sin(0.005) * 1.0001 + 1e-6);
        const double zero[9] = {0};
        CHECK(spectral_norm3(zero) <= 1e-11);
        const double diag[9] = {3, 0, 0, 0, -7, 0, 0, 0, 2};
        CHECK(std::fabs(spectral_norm3(diag) - 7.0) < 1e-4);
    }
}

// ---- cell-list reuse

static CellCamera cell_camera(double yaw, float px, float py, float pz, double drift = 0.0, uint64_t gen = 1)
{
    CellCamera c;
    c.view = yaw_view(yaw, px, py, pz);
    c.e1 = 6.2354f;
    c.e2 = 0.57735f;
    c.W = 1920;
    c.H = 1080;
    c.drift = drift;
    c.scene_gen = gen;
    return c;
}

static void test_cell_motion_bounds_direction_change()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int it = 0; it < 2000; it++) {
        const double yaw0 = 3.0 * u(rng), dyaw = 0.05 * u(rng);
        const CellCamera a = cell_camera(yaw0, 0, 0, 0), b = cell_camera(yaw0 + dyaw, (float)(0.3 * u(rng)), (float)(0.3 * u(rng)), (float)(0.3 * u(rng)));
        const CellBudget m = cell_motion(a, b);
        CHECK(m.theta < 1.0e29f && m.delta < 1.0e29f);
        // the same pixel's unit direction under both matrices: |d' - d| <= theta
        for (int k = 0; k < 8; k++) {
            const double p[3] = {6.2354 * u(rng), 0.57735 * u(rng), 1.0};
            double d0[3], d1[3], n0 = 0, n1 = 0;
            for (int r = 0; r < 3; r++) {
                d0[r] = a.view.rot[3 * r] * p[0] + a.view.rot[3 * r + 1] * p[1] + a.view.rot[3 * r + 2] * p[2];
                d1[r] = b.view.rot[3 * r] * p[0] + b.view.rot[3 * r + 1] * p[1] + b.view.rot[3 * r + 2] * p[2];
                n0 += d0[r] * d0[r];
                n1 += d1[r] * d1[r];
            }
            double e = 0;
            for (int r = 0; r < 3; r++) {
                const double x = d1[r] / std::sqrt(n1) - d0[r] / std::sqrt(n0);
                e += x * x;
            }
            CHECK(std::sqrt(e) <= m.theta);
        }
        const double dp = std::sqrt((double)b.view.pos[0] * b.view.pos[0] + (double)b.view.pos[1] * b.view.pos[1] + (double)b.view.pos[2] * b.view.pos[2]);
        CHECK(m.delta >= (float)dp);
    }
    // identical cameras use nothing; anything else the lists depend on makes them unusable
    const CellCamera a = cell_camera(3.14159265, 1, 2, 3, 5.0);
    CellBudget m = cell_motion(a, a);
    CHECK(m.theta == 0.0f && m.delta == 0.0f);
    CellCamera b = a;
    b.scene_gen = 2;
    CHECK(cell_motion(a, b).theta > 1.0e29f);
    b = a;
    b.e1 = 6.0f;
    CHECK(cell_motion(a, b).delta > 1.0e29f);
    b = a;
    b.drift = 4.0; // drift only grows
    CHECK(cell_motion(a, b).delta > 1.0e29f);
    b = a;
    b.drift = 5.5;
    CHECK(cell_motion(a, b).delta >= 0.5f && cell_motion(a, b).delta < 0.51f);
    // a matrix that is not a rotation gets no rotation budget unless it is the very same matrix
    CellCamera s = a;
    for (float& x : s.view.rot) x *= 2.0f;
    CHECK(cell_motion(s, s).theta == 0.0f);
    CellCamera s2 = s;
    s2.view.rot[0] += 1.0e-3f;
    CHECK(cell_motion(s, s2).theta > 1.0e29f);
    b = a;
    b.view.rot[4] = std::nanf("");
    CHECK(cell_motion(a, b).theta > 1.0e29f);
}

// The inequality behind the grown margin, checked numerically in double: a point within R of the centre on a ray of
// the moved camera never lies further behind an old pyramid plane than margin' = R + delta + theta (|O'| + R).
static void test_grown_margin_inequality()
{
    std::mt19937 rng(11);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    auto unit = [&](double v[3]) {
        double n;
        do {
            for (int k = 0; k < 3; k++) v[k] = u(rng);
            n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        } while (n < 0.1 || n > 1.0);
        for (int k = 0; k < 3; k++) v[k] /= n;
    };
    for (int it = 0; it < 200000; it++) {
        double n[3], d[3], w[3], w2[3], w3[3];
        unit(n);
        unit(d);
        const double nd = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
        if (nd < 0) {
            for (int k = 0; k < 3; k++) d[k] = -d[k]; // the old ray lies on the inner side of the plane
        }
        const double theta = 0.05 * std::fabs(u(rng)), delta = 2.0 * std::fabs(u(rng)), R = 3.0 * std::fabs(u(rng));
        // d': a unit vector within chord theta of d
        unit(w);
        double dp[3], len = 0;
        const double sc = 0.999 * theta * std::fabs(u(rng));
        for (int k = 0; k < 3; k++) dp[k] = d[k] + sc * w[k];
        for (int k = 0; k < 3; k++) len += dp[k] * dp[k];
        for (int k = 0; k < 3; k++) dp[k] /= std::sqrt(len);
        double chord = 0;
        for (int k = 0; k < 3; k++) chord += (dp[k] - d[k]) * (dp[k] - d[k]);
        if (std::sqrt(chord) > theta) continue;
        // o' within delta of o (= 0); p on the new ray; c within R of p
        unit(w2);
        unit(w3);
        const double t = 200.0 * std::fabs(u(rng)), so = delta * std::fabs(u(rng)), sr = R * std::fabs(u(rng));
        double o2[3], c[3];
        for (int k = 0; k < 3; k++) {
            o2[k] = so * w2[k];
            c[k] = o2[k] + t * dp[k] + sr * w3[k];
        }
        double O2 = 0;
        for (int k = 0; k < 3; k++) O2 += (o2[k] - c[k]) * (o2[k] - c[k]);
        O2 = std::sqrt(O2);
        const double lhs = n[0] * c[0] + n[1] * c[1] + n[2] * c[2]; // n . (c - o)
        CHECK(lhs >= -(R + delta + theta * (O2 + R)) - 1e-9);
    }
}

static const bool kReady[2] = {true, true};

static void test_cell_cache_policy()
{
    CellKey key;
    key.W = 1920;
    key.H = 1080;
    key.rows = 1080;
    key.lw = 4;
    key.nsub = 2;
    key.ns = 65536;
    key.cap = 1280;
    const double cw = 0.208, ch = 0.137; // a cell's extent on the view plane (config 5: 32 x 128 pixels)
    auto covered = [&](const CellCachePolicy& pol, const std::vector<std::pair<CellCamera, CellBudget>>& built, int slot, const CellCamera& cam) {
        (void)pol;
        const CellBudget m = cell_motion(built[(size_t)slot].first, cam);
        return m.theta <= built[(size_t)slot].second.theta && m.delta <= built[(size_t)slot].second.delta;
    };
    // static view: one build, then hits for ever; zero budgets (the lists are exact)
    {
        CellCachePolicy pol;
        std::vector<std::pair<CellCamera, CellBudget>> built(2);
        int builds = 0, hits = 0;
        for (int f = 0; f < 100; f++) {
            const CellCamera cam = cell_camera(3.14159265, 0, 0, 0);
            const CellCachePolicy::Decision d = pol.decide(key, cam, cw, ch, kReady);
            if (d.action == CellCachePolicy::kBuild) {
                builds++;
                CHECK(d.budget.theta == 0.0f && d.budget.delta == 0.0f);
                built[(size_t)d.slot] = {cam, d.budget};
            } else {
                CHECK(d.action == CellCachePolicy::kUse && !d.prefetch);
                hits++;
            }
        }
        CHECK(builds == 1 && hits == 99);
    }
    // a camera that turns slowly and drifts: after the first frames every launch is served by lists that cover it (the
    // decision is re-checked here against what the slots were built with), rebuilt ahead of time
    for (double step : {1.0e-4, 5.0e-4, 1.0e-3}) {
        CellCachePolicy pol;
        std::vector<std::pair<CellCamera, CellBudget>> built(2);
        int builds = 0, hits = 0, prefetches = 0, per_frame = 0;
        for (int f = 0; f < 600; f++) {
            const CellCamera cam = cell_camera(3.14159265 + step * f, (float)(0.01 * f), 0, 0, 0.002 * f);
            const CellCachePolicy::Decision d = pol.decide(key, cam, cw, ch, kReady);
            if (d.action == CellCachePolicy::kBuild) {
                builds++;
                built[(size_t)d.slot] = {cam, d.budget};
            } else if (d.action == CellCachePolicy::kUse) {
                hits++;
                CHECK(covered(pol, built, d.slot, cam));
            } else {
                per_frame++;
            }
            if (d.prefetch) {
                prefetches++;
                CHECK(d.action == CellCachePolicy::kUse && d.prefetch_slot != d.slot);
                built[(size_t)d.prefetch_slot] = {cam, d.prefetch_budget};
            }
        }
        CHECK(per_frame == 0);
        CHECK(builds <= 3);               // the start-up only (no step known yet; then the first real budget)
        CHECK(hits >= 595 && prefetches >= 600 / 10 && prefetches <= 600 / 2);
    }
    // a camera too fast (a quarter of a cell in under four frames): per-frame binning, no cache traffic
    {
        CellCachePolicy pol;
        int per_frame = 0;
        for (int f = 0; f < 50; f++) {
            const CellCachePolicy::Decision d = pol.decide(key, cell_camera(3.14159265 + 0.01 * f, 0, 0, 0), cw, ch, kReady);
            per_frame += d.action == CellCachePolicy::kPerFrame;
            CHECK(!d.prefetch);
        }
        CHECK(per_frame >= 48);
    }
    // ... and when it comes to rest the lists are built again
    {
        CellCachePolicy pol;
        for (int f = 0; f < 20; f++) CHECK(pol.decide(key, cell_camera(3.14159265 + 0.01 * f, 0, 0, 0), cw, ch, kReady).action == CellCachePolicy::kPerFrame || f == 0);
        int builds = 0, hits = 0;
        for (int f = 0; f < 20; f++) {
            const CellCachePolicy::Decision d = pol.decide(key, cell_camera(3.14159265 + 0.2, 0, 0, 0), cw, ch, kReady);
            builds += d.action == CellCachePolicy::kBuild;
            hits += d.action == CellCachePolicy::kUse;
        }
        CHECK(builds == 1 && hits == 18);
    }
    // a scene edit, another grid, an explicit invalidate: never served from the old lists
    {
        CellCachePolicy pol;
        const CellCamera cam = cell_camera(3.14159265, 0, 0, 0);
        CHECK(pol.decide(key, cam, cw, ch, kReady).action == CellCachePolicy::kBuild);
        CHECK(pol.decide(key, cam, cw, ch, kReady).action == CellCachePolicy::kUse);
        CellCamera edited = cam;
        edited.scene_gen = 2;
        CHECK(pol.decide(key, edited, cw, ch, kReady).action == CellCachePolicy::kBuild); // (an edit is not a fast camera: rebuilt at once)
        CHECK(pol.decide(key, edited, cw, ch, kReady).action == CellCachePolicy::kUse);
        CellKey other = key;
        other.rows = 540;
        CHECK(pol.decide(other, edited, cw, ch, kReady).action == CellCachePolicy::kBuild);
        CHECK(pol.decide(other, edited, cw, ch, kReady).action == CellCachePolicy::kUse);
        CHECK(pol.decide(key, edited, cw, ch, kReady).action == CellCachePolicy::kUse);   // the first grid's lists are in the other slot still
        pol.invalidate();
        CHECK(pol.decide(key, edited, cw, ch, kReady).action == CellCachePolicy::kBuild);
    }
    // lists being built ahead of time are not switched to while finished ones still cover the camera
    {
        CellCachePolicy pol;
        int switched_early = 0, prefetched_at = -1, used_slot = -1;
        bool ready[2] = {true, true};
        for (int f = 0; f < 40; f++) {
            const CellCamera cam = cell_camera(3.14159265 + 5.0e-4 * f, 0, 0, 0);
            const CellCachePolicy::Decision d = pol.decide(key, cam, cw, ch, ready);
            if (d.action == CellCachePolicy::kUse && prefetched_at >= 0 && f <= prefetched_at + 2 && d.slot != used_slot) switched_early++;
            if (d.action != CellCachePolicy::kPerFrame) used_slot = d.slot;
            if (d.prefetch && prefetched_at < 0) {
                prefetched_at = f;
                ready[d.prefetch_slot] = false;           // its build takes three frames
            }
            if (prefetched_at >= 0 && f == prefetched_at + 2) ready[0] = ready[1] = true;
        }
        CHECK(prefetched_at >= 0 && switched_early == 0);
    }
    // capacity feedback: grows past what was seen once the lists have less than a fifth to spare, never shrinks
    CHECK(cell_capacity_wanted(100, 1509, 0) == 0);
    CHECK(cell_capacity_wanted(1300, 1509, 0) == 2048);
    CHECK(cell_capacity_wanted(5000, 1509, 1792) == 7680);
    CHECK(cell_capacity_wanted(100, 1509, 1792) == 1792);
    {
        const TileRequest q = request(1920, 1080, 1080, 65536, true);
        const TileShape t = plan_tiles(q);
        CHECK(plan_cells(t, 65536, q.aspect, 0, 4096).cap == 4096);
        CHECK(plan_cells(t, 65536, q.aspect, 0, 1u << 20).cap == 65536);   // never more than the scene
        CHECK(plan_cells(t, 65536, q.aspect, 7, 4096).cap == 7);            // an explicit capacity is taken as is
    }
    // still_only (scenes under 2048 spheres): lists only while camera and scene rest
    {
        CellCachePolicy pol;
        std::vector<int> actions;
        auto act = [&](const CellCamera& c) {
            const CellCachePolicy::Decision d = pol.decide(key, c, cw, ch, kReady, true);
            CHECK(!d.prefetch);
            if (d.action == CellCachePolicy::kBuild) CHECK(d.budget.theta == 0.0f && d.budget.delta == 0.0f);
            return (int)d.action;
        };
        const CellCamera rest = cell_camera(3.14159265, 0, 0, 0);
        CHECK(act(rest) == CellCachePolicy::kSkip);    // nothing before it to compare with
        CHECK(act(rest) == CellCachePolicy::kSkip);    // one launch without motion
        CHECK(act(rest) == CellCachePolicy::kBuild);   // two: build exact lists
        for (int f = 0; f < 50; f++) CHECK(act(rest) == CellCachePolicy::kUse);
        for (int f = 1; f <= 30; f++) CHECK(act(cell_camera(3.14159265 + 1.0e-4 * f, 0, 0, 0)) == CellCachePolicy::kSkip); // moving: whole-scene staging
        const CellCamera there = cell_camera(3.14159265 + 30.0e-4, 0, 0, 0);
        CHECK(act(there) == CellCachePolicy::kSkip && act(there) == CellCachePolicy::kBuild && act(there) == CellCachePolicy::kUse);
        CHECK(act(rest) == CellCachePolicy::kUse);     // back where the other set was built: still there
        CellCamera moved_scene = rest;
        moved_scene.drift = 0.5;                       // physics moved the spheres: the lists are not exact any more
        CHECK(act(moved_scene) == CellCachePolicy::kSkip);
    }
    // physics: spheres that move (drift) use the position budget like a moving camera
    {
        CellCachePolicy pol;
        std::vector<std::pair<CellCamera, CellBudget>> built(2);
        int misses = 0;
        for (int f = 0; f < 300; f++) {
            const CellCamera cam = cell_camera(3.14159265, 0, 0, 0, 0.06 * f);
            const CellCachePolicy::Decision d = pol.decide(key, cam, cw, ch, kReady);
            if (d.action == CellCachePolicy::kBuild) {
                built[(size_t)d.slot] = {cam, d.budget};
                misses += f > 2;
            } else if (d.action == CellCachePolicy::kUse) {
                CHECK(covered(pol, built, d.slot, cam));
            }
            if (d.prefetch) built[(size_t)d.prefetch_slot] = {cam, d.prefetch_budget};
        }
        CHECK(misses == 0);
    }
}

static void test_view_density()
{
    ViewDensity v;
    CHECK(!v.dense() && v.report_from() == ViewDensity::kReportSparse);
    CHECK(!v.observe(0) && !v.observe(9) && !v.observe(27) && !v.dense());   // the default view of config 2: 9 at most
    CHECK(v.observe(28) && v.dense() && v.report_from() == ViewDensity::kLightDense);
    CHECK(!v.observe(95));                                                    // a sparse-plan value that was in flight: stays dense
    CHECK(!v.observe(0) && !v.observe(0));                                    // two calm epochs are not enough ...
    CHECK(!v.observe(14) && v.dense());                                       // ... and a long list starts the count again
    CHECK(!v.observe(11) && !v.observe(0) && v.observe(3) && !v.dense());     // three in a row: sparse again
    CHECK(!v.observe(8) && !v.dense());                                       // a dense-plan value that was in flight: below kHeavy
    v.observe(40);
    v.reset();
    CHECK(!v.dense());
    // the dense plan for a scene that is sparse by its numbers: 2 sub-tiles, refinement; the sparse plan is untouched by the flag's absence
    TileRequest q = request(1920, 1080, 1080, 1025, true);
    const TileShape sparse = plan_tiles(q);
    q.view_dense = true;
    const TileShape dense = plan_tiles(q);
    CHECK(sparse.nsub == 5 && !sparse.refine && !sparse.dense);
    CHECK(dense.nsub == 2 && dense.refine && dense.dense && dense.mw * dense.mh == 512);
}

// The direction order: a permutation with its inverse; spheres of a narrow cone are neighbours in it (few 8-sphere lines).
static void test_direction_order()
{
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const uint32_t n = 20000;
    std::vector<float> c(4 * n);
    for (uint32_t k = 0; k < n; k++) {
        const float d = 40.0f + 80.0f * (u(rng) + 1.0f), tx = 5.9f * u(rng), ty = 0.55f * u(rng), s = d / std::sqrt(1.0f + tx * tx + ty * ty);
        c[4 * k] = s * tx;
        c[4 * k + 1] = s * ty;
        c[4 * k + 2] = s;
        c[4 * k + 3] = 1.0f;
    }
    const float origin[3] = {0, 0, 0};
    std::vector<uint32_t> order, pos_of;
    direction_order(c.data(), 4, n, origin, order, pos_of);
    CHECK(order.size() == n && pos_of.size() == n);
    std::vector<int> seen(n, 0);
    for (uint32_t p = 0; p < n; p++) {
        CHECK(order[p] < n);
        seen[order[p]]++;
        CHECK(pos_of[order[p]] == p);
    }
    CHECK(std::all_of(seen.begin(), seen.end(), [](int x) { return x == 1; }));
    // the spheres inside a 1/20 x 1/10 window of the view: lines of 8 touched in creation order vs in the direction order
    size_t lines_creation = 0, lines_sorted = 0, members = 0;
    for (int wx = 0; wx < 20; wx += 3) {
        for (int wy = 0; wy < 10; wy += 3) {
            std::vector<uint32_t> a, b;
            for (uint32_t k = 0; k < n; k++) {
                const float tx = c[4 * k] / c[4 * k + 2], ty = c[4 * k + 1] / c[4 * k + 2];
                if (tx >= -5.9f + 0.59f * wx && tx < -5.9f + 0.59f * (wx + 1) && ty >= -0.55f + 0.11f * wy && ty < -0.55f + 0.11f * (wy + 1)) {
                    a.push_back(k / 8);
                    b.push_back(pos_of[k] / 8);
                }
            }
            members += a.size();
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            lines_creation += (size_t)(std::unique(a.begin(), a.end()) - a.begin());
            lines_sorted += (size_t)(std::unique(b.begin(), b.end()) - b.begin());
        }
    }
    CHECK(members > 500 && lines_sorted * 3 < lines_creation); // (measured on config 5's scene: 9.3 MB of lines against 2.2)
    // degenerate inputs: coincident with the origin, NaN; one sphere; none
    std::vector<float> d = {0, 0, 0, 1, std::nanf(""), 1, 2, 1, 3, 3, 3, 1};
    direction_order(d.data(), 4, 3, origin, order, pos_of);
    CHECK(order.size() == 3 && pos_of[order[0]] == 0 && pos_of[order[1]] == 1 && pos_of[order[2]] == 2);
    direction_order(d.data(), 4, 1, origin, order, pos_of);
    CHECK(order.size() == 1 && order[0] == 0 && pos_of[0] == 0);
    direction_order(d.data(), 4, 0, origin, order, pos_of);
    CHECK(order.empty() && pos_of.empty());
}


// EdgeBasis: the side planes of the culling pyramids from three per-frame vectors.  The device evaluates t P + Q in fp32
// (-ffp-contract=off: a rounded product, a rounded sum); this test does the same in float and compares with the plane's normal
// in double.  Beside it, the formula the kernels used before -- the fp32 cross product of the two corner directions -- on the
// same edges: at 8K, for a tile at the frame's edge, it is off by more than the half pixel the pyramid is grown by.
static void rotation(double pitch, double yaw, double roll, float m[16])
{
    const double cx = std::cos(pitch), sx = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw), cz = std::cos(roll), sz = std::sin(roll);
    const double r[9] = {cy * cz + sy * sx * sz, -cy * sz + sy * sx * cz, sy * cx, cx * sz, cx * cz, -sx,
                         -sy * cz + cy * sx * sz, sy * sz + cy * sx * cz, cy * cx};
    for (int k = 0; k < 16; k++) m[k] = 0.0f;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) m[4 * i + j] = (float)r[3 * i + j];
    }
    m[15] = 1.0f;
}

static void test_edge_basis()
{
    std::mt19937 rng(99);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    double worst_new = 0.0, worst_old = 0.0, worst_contain = 0.0;
    for (int it = 0; it < 400; it++) {
        const uint64_t H = it % 3 == 0 ? 1080 : (it % 3 == 1 ? 2160 : 4320), W = H * 16 / 9;
        const float e1 = 0.57735f * (float)H / 100.0f, e2 = 0.57735f;
        float m[16];
        rotation(0.4 * u(rng), 3.14159265 + 0.5 * u(rng), 0.4 * u(rng), m);
        const EdgeBasis e = edge_basis(m, e1, e2, W, H);
        auto dir = [&](double cx, double cy, double* w) {
            for (int k = 0; k < 3; k++) w[k] = (double)m[4 * k] * (cx * e1) + (double)m[4 * k + 1] * (cy * e2) + (double)m[4 * k + 2];
        };
        auto dirf = [&](float cx, float cy, float* w) { // view_dir of the kernels, fp32
            const float vx = cx * e1, vy = cy * e2;
            for (int k = 0; k < 3; k++) w[k] = m[4 * k] * vx + m[4 * k + 1] * vy + m[4 * k + 2];
        };
        auto angle = [](const double* a, const double* b) { // between lines
            const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            return std::sqrt((c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) / ((a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * (b[0] * b[0] + b[1] * b[1] + b[2] * b[2])));
        };
        // a 16-column, 64-row tile at a random place, often at the frame's left or right edge
        const uint64_t col0 = it % 2 ? (uint64_t)((W - 17) * (0.5 + 0.5 * u(rng))) : (it % 4 ? 0 : W - 16);
        const uint64_t row0 = (uint64_t)((H - 65) * (0.5 + 0.5 * u(rng)));
        const float fW = (float)W, fH = (float)H;
        const float x0 = (2.0f * (float)col0 - 1.0f - fW) / fW, x1 = (2.0f * (float)(col0 + 16) - 1.0f - fW) / fW;
        const float y0 = (fH - 2.0f * (float)row0 + 1.0f) / fH, y1 = (fH - 2.0f * (float)(row0 + 64) + 1.0f) / fH;
        for (int k = 0; k < 4; k++) {
            const bool row_edge = (k & 1) == 0;
            const float t = row_edge ? (k == 0 ? y0 : y1) : (k == 1 ? x1 : x0);
            const float* pb = row_edge ? e.up_p : e.right_p;
            const float* q = row_edge ? e.up_q : e.right_q;
            float nf[3];
            for (int i = 0; i < 3; i++) {
                volatile float prod = t * pb[i]; // (volatile: no contraction into an FMA, as on the device)
                nf[i] = (k < 2 ? -1.0f : 1.0f) * (prod + q[i]); // inward: down from the top, left from the right, up from the bottom, right from the left
            }
            {
                // ... and inward it is: positive on the rectangle's central direction
                double wc[3];
                dir(0.5 * ((double)x0 + x1), 0.5 * ((double)y0 + y1), wc);
                CHECK((double)nf[0] * wc[0] + (double)nf[1] * wc[1] + (double)nf[2] * wc[2] > 0.0);
            }
            const double hyp2 = (double)t * t * e.pp + (row_edge ? e.qrqr : e.qcqc);
            CHECK((double)nf[0] * nf[0] + (double)nf[1] * nf[1] + (double)nf[2] * nf[2] >= 0.25 * hyp2); // a camera matrix never cancels
            // the exact plane: through the two corner directions, in double
            const double xa = (k == 1 || k == 2) ? x1 : x0, ya = k >= 2 ? y1 : y0, xb = ((k + 1) % 4 == 1 || (k + 1) % 4 == 2) ? x1 : x0, yb = (k + 1) % 4 >= 2 ? y1 : y0;
            double wa[3], wb[3];
            dir(xa, ya, wa);
            dir(xb, yb, wb);
            const double nt[3] = {wa[1] * wb[2] - wa[2] * wb[1], wa[2] * wb[0] - wa[0] * wb[2], wa[0] * wb[1] - wa[1] * wb[0]};
            const double nd[3] = {nf[0], nf[1], nf[2]};
            worst_new = std::max(worst_new, angle(nd, nt));
            // every pixel direction along the edge lies in the plane
            for (int sidx = 0; sidx <= 4; sidx++) {
                double wm[3];
                dir(xa + (xb - xa) * sidx / 4.0, ya + (yb - ya) * sidx / 4.0, wm);
                const double dn = (nd[0] * wm[0] + nd[1] * wm[1] + nd[2] * wm[2]) /
                                  std::sqrt((nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]) * (wm[0] * wm[0] + wm[1] * wm[1] + wm[2] * wm[2]));
                worst_contain = std::max(worst_contain, std::fabs(dn));
            }
            // the former formula: cross product of the fp32 corner directions, in fp32
            float fa[3], fb[3];
            dirf((float)xa, (float)ya, fa);
            dirf((float)xb, (float)yb, fb);
            volatile float p0 = fa[1] * fb[2], p1 = fa[2] * fb[1], p2 = fa[2] * fb[0], p3 = fa[0] * fb[2], p4 = fa[0] * fb[1], p5 = fa[1] * fb[0];
            const double no[3] = {(double)(float)(p0 - p1), (double)(float)(p2 - p3), (double)(float)(p4 - p5)};
            worst_old = std::max(worst_old, angle(no, nt));
        }
    }
    std::printf("edge planes: worst direction error %.2e rad (t P + Q), %.2e rad (fp32 cross product of corner directions); worst |n . w| %.2e\n",
                worst_new, worst_old, worst_contain);
    CHECK(worst_new < 6.0e-7);      // the kernel's comment says 4e-7 from the roundings, and the edge coordinates here are the same floats
    CHECK(worst_contain < 6.0e-7);
    CHECK(worst_old > 2.0e-5);      // why the formula changed (half a pixel at the 8K frame's edge is 5e-6 rad)
    // the camera plane: every pixel ray of the frame in front of it, for every frame size the reference's camera produces
    for (uint64_t H : {180ull, 1080ull, 4320ull, 17280ull}) {
        const uint64_t W = H * 16 / 9;
        float m[16];
        rotation(0.3, 2.0, -0.2, m);
        const float e1 = 0.57735f * (float)H / 100.0f, e2 = 0.57735f;
        const EdgeBasis e = edge_basis(m, e1, e2, W, H);
        CHECK(std::fabs((double)e.fwd[0] * e.fwd[0] + (double)e.fwd[1] * e.fwd[1] + (double)e.fwd[2] * e.fwd[2] - 1.0) < 1e-6);
        for (int i = 0; i < 4; i++) {
            const double cx = (i & 1) ? 1.0 : -1.0, cy = (i & 2) ? 1.0 : -1.0;
            double w[3];
            for (int k = 0; k < 3; k++) w[k] = (double)m[4 * k] * (cx * e1) + (double)m[4 * k + 1] * (cy * e2) + (double)m[4 * k + 2];
            CHECK(e.fwd[0] * w[0] + e.fwd[1] * w[1] + e.fwd[2] * w[2] > 0.0);
        }
    }
    {
        // a frame so wide that its corner rays are within 1e-4 of perpendicular to the axis: no camera plane
        float m[16];
        rotation(0.0, 0.0, 0.0, m);
        const EdgeBasis e = edge_basis(m, 3.0e4f, 0.5f, 1920, 1080);
        CHECK(e.fwd[0] == 0.0f && e.fwd[1] == 0.0f && e.fwd[2] == 0.0f && e.pp > 0.0f);
    }
    // degenerate parameters: never a plane that culls
    float z[16] = {0};
    EdgeBasis d = edge_basis(z, 1.0f, 1.0f, 640, 360);
    CHECK(std::isinf(d.pp) && d.up_p[0] == 0.0f && d.up_q[0] == 0.0f && d.fwd[2] == 0.0f);
    z[0] = std::nanf("");
    d = edge_basis(z, 1.0f, 1.0f, 640, 360);
    CHECK(std::isinf(d.pp) && std::isinf(d.qrqr) && std::isinf(d.qcqc) && d.up_p[0] == 0.0f && d.fwd[0] == 0.0f); // refusal test fails for every t
    float big[16];
    rotation(0.1, 0.2, 0.3, big);
    d = edge_basis(big, 3.0e38f, 3.0e38f, 640, 360);
    CHECK(std::isinf(d.pp) && d.up_q[0] == 0.0f);
    float flat[16];
    rotation(0.1, 0.2, 0.3, flat);
    flat[2] = flat[6] = flat[10] = 0.0f; // no third column: every direction in one plane
    d = edge_basis(flat, 1.0f, 1.0f, 640, 360);
    CHECK(std::isinf(d.pp) && d.right_q[0] == 0.0f);
    // a sheared matrix (column 1 nearly parallel to column 2): row edges near cy = -1/e2 cancel and are refused by the test the
    // device applies, others are kept
    float sh[16];
    rotation(0.0, 0.0, 0.0, sh);
    sh[1] = sh[2];
    sh[5] = sh[6] + 1.0e-3f;
    sh[9] = sh[10];
    d = edge_basis(sh, 1.0f, 1.0f, 640, 360);
    {
        const float t = -1.0f;
        float nf[3];
        for (int i = 0; i < 3; i++) nf[i] = t * d.up_p[i] + d.up_q[i];
        const double len2 = (double)nf[0] * nf[0] + (double)nf[1] * nf[1] + (double)nf[2] * nf[2];
        CHECK(!(len2 >= 0.25 * ((double)t * t * d.pp + d.qrqr)));
    }
}

int main()
{
    test_edge_basis();
    test_direction_order();
    test_view_density();
    test_tile_shapes();
    test_cell_grid();
    test_xcd_order();
    test_dispatch_order_static_view();
    test_dispatch_order_moving_views();
    test_cell_motion_bounds_direction_change();
    test_grown_margin_inequality();
    test_cell_cache_policy();
    static_assert(kResidentPerCU == RTX_WAVES_PER_EU, "one constant");
    if (g_failed) {
        std::printf("%d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("all host planning tests passed\n");
    return 0;
}
