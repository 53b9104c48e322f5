// TEST INFRASTRUCTURE ONLY (built into oracle/_ref/ref_run by build_ref_shim.sh; nothing in alvaar_amd/, include/ or bench.py's timed
// region runs it).  ONE run of the reference's own System (ref_shim_system.cpp: ref_system_*) over a stream of frames, in a process of
// its own that is nothing but the reference: no interpreter, no other library's allocations, address-space randomisation off.
//
// Why: Ceres keeps its parameter blocks ordered by ADDRESS, so the order of its reductions -- hence the last bits of a local BA, hence,
// a few hundred frames later, a discrete decision -- follows the heap layout.  Inside the test's Python process two runs on the same
// frames differ from each other (tools/ref_determinism_probe.py), and so do runs in fresh interpreters even without ASLR (byte-code
// caches written or not, environment, path lengths all move the heap).  Here the heap holds the reference's own allocations only, in
// program order, from a fixed base: the run is a function of the job file.  tests/ref_runner.py writes the job and parses the records.
//
//   ref_run <job> <records>     (re-executes itself under personality(ADDR_NO_RANDOMIZE) when it was started without it)
//
// job (little endian): int32 magic 'ARJ1', w, h, cell, clahe, n_base, n_steps; double fx, fy, cx, cy, k1, k2, p1, p2;
//                      int32 base_index[n_steps]; int32 reset_before[n_steps]; double timestamp[n_steps]; uint8 gray[n_base][h][w]
//                      (a frame is RGBA = (g, g, g, 255): synth.gray_to_rgba)
// records, per step:   int32 status; double pose7[7]; float pose16[16]; int32 state[16];
//                      int32 n_kp; int32 id[n]; float px[n][2]; float unpx[n][2]; uint8 is3d[n]; uint8 has_desc[n];
//                      int32 n_kf; int32 kfid[n]; int32 n_mp; int32 id[n]; double xyz[n][3]; int32 flags[n][5]; double inv_depth[n]; uint8 desc[n][32]
// after the last step: int32 n_kf, then per keyframe of the final map (ascending id):
//                      int32 kfid; double pose7[7]; int32 info[6]; int32 n; int32 id[n]; float px[n][2]; uint8 is3d[n]; int32 n_cov; int32 cov[n_cov][2]
#include <sys/personality.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
void ref_freeze_clock(int on);
void *ref_system_create(int w, int h, double fx, double fy, double cx, double cy, double k1, double k2, double p1, double p2, int cellSize, int claheEnabled,
                        int doRandom);
void ref_system_destroy(void *p);
void ref_system_reset(void *p);
int ref_system_find_camera_pose(void *p, const uint8_t *rgba, double timestamp, float *pose16, double *pose7);
void ref_system_state(void *p, int *out16);
int ref_system_frame_keypoints(void *p, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *has_desc);
int ref_system_keyframe_ids(void *p, int cap, int *ids);
int ref_system_map_points(void *p, int cap, int *ids, double *xyz, int *flags5, double *inv_depth, uint8_t *desc32);
int ref_system_keyframe(void *p, int kfid, double *pose7, int *info6, int cap, int *ids, float *px, uint8_t *is3d);
int ref_system_covisibility(void *p, int kfid, int cap, int *pairs);
}

static void need(bool ok, const char *what) {
    if (!ok) {
        fprintf(stderr, "ref_run: %s\n", what);
        exit(2);
    }
}

int main(int argc, char **argv) {
    need(argc == 3, "usage: ref_run <job> <records>");
    const int pers = personality(0xffffffff);
    if (!(pers & ADDR_NO_RANDOMIZE)) {
        need(personality(pers | ADDR_NO_RANDOMIZE) >= 0, "personality(ADDR_NO_RANDOMIZE) refused");
        execv("/proc/self/exe", argv);
        need(false, "execv(/proc/self/exe) failed");
    }
    FILE *f = fopen(argv[1], "rb");
    need(f != nullptr, "cannot open the job file");
    int32_t hd[7];
    double cam[8];
    need(fread(hd, 4, 7, f) == 7 && hd[0] == 0x314a5241 && fread(cam, 8, 8, f) == 8, "bad job header");
    const int w = hd[1], h = hd[2], cell = hd[3], clahe = hd[4], n_base = hd[5], n_steps = hd[6];
    std::vector<int32_t> idx(n_steps), rst(n_steps);
    std::vector<double> ts(n_steps);
    need(fread(idx.data(), 4, n_steps, f) == (size_t) n_steps && fread(rst.data(), 4, n_steps, f) == (size_t) n_steps &&
             fread(ts.data(), 8, n_steps, f) == (size_t) n_steps, "bad job tables");
    const size_t P = (size_t) w * h;
    std::vector<uint8_t> gray(P * n_base), rgba(P * 4);
    need(fread(gray.data(), 1, gray.size(), f) == gray.size(), "bad job frames");
    fclose(f);
    FILE *o = fopen(argv[2], "wb");
    need(o != nullptr, "cannot open the records file");
    constexpr int CAP_KP = 16384, CAP_MP = 65536;
    std::vector<int32_t> kid(CAP_KP), kfid(256), mid(CAP_MP), mfl((size_t) CAP_MP * 5);
    std::vector<float> kpx((size_t) CAP_KP * 2), kun((size_t) CAP_KP * 2);
    std::vector<uint8_t> k3(CAP_KP), khd(CAP_KP), mdesc((size_t) CAP_MP * 32);
    std::vector<double> mxyz((size_t) CAP_MP * 3), minv(CAP_MP);
    ref_freeze_clock(1);   // Ceres' wall-clock caps never fire
    void *sys = ref_system_create(w, h, cam[0], cam[1], cam[2], cam[3], cam[4], cam[5], cam[6], cam[7], cell, clahe, 0);
    for (int k = 0; k < n_steps; k++) {
        need(idx[k] >= 0 && idx[k] < n_base, "base index out of range");
        const uint8_t *g = gray.data() + P * idx[k];
        for (size_t i = 0; i < P; i++) {
            rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = g[i];
            rgba[4 * i + 3] = 255;
        }
        if (rst[k]) ref_system_reset(sys);
        float pose16[16] = {0};
        double pose7[7] = {0};
        int32_t state[16] = {0};
        const int32_t st = ref_system_find_camera_pose(sys, rgba.data(), ts[k], pose16, pose7);
        ref_system_state(sys, state);
        const int32_t nk = ref_system_frame_keypoints(sys, CAP_KP, kid.data(), kpx.data(), kun.data(), k3.data(), khd.data());
        const int32_t nf = ref_system_keyframe_ids(sys, 256, kfid.data());
        const int32_t nm = ref_system_map_points(sys, CAP_MP, mid.data(), mxyz.data(), mfl.data(), minv.data(), mdesc.data());
        need(nk >= 0 && nk <= CAP_KP && nf >= 0 && nf <= 256 && nm >= 0 && nm <= CAP_MP, "a table outgrew its capacity");
        fwrite(&st, 4, 1, o); fwrite(pose7, 8, 7, o); fwrite(pose16, 4, 16, o); fwrite(state, 4, 16, o);
        fwrite(&nk, 4, 1, o); fwrite(kid.data(), 4, nk, o); fwrite(kpx.data(), 4, 2 * (size_t) nk, o); fwrite(kun.data(), 4, 2 * (size_t) nk, o);
        fwrite(k3.data(), 1, nk, o); fwrite(khd.data(), 1, nk, o);
        fwrite(&nf, 4, 1, o); fwrite(kfid.data(), 4, nf, o);
        fwrite(&nm, 4, 1, o); fwrite(mid.data(), 4, nm, o); fwrite(mxyz.data(), 8, 3 * (size_t) nm, o); fwrite(mfl.data(), 4, 5 * (size_t) nm, o);
        fwrite(minv.data(), 8, nm, o); fwrite(mdesc.data(), 1, 32 * (size_t) nm, o);
    }
    {   // the final map's keyframes (sysdiff.compare_keyframes)
        const int32_t nf = ref_system_keyframe_ids(sys, 256, kfid.data());
        fwrite(&nf, 4, 1, o);
        std::vector<int32_t> cov(512);
        for (int i = 0; i < nf; i++) {
            double pose7[7];
            int32_t info[6];
            const int32_t n = ref_system_keyframe(sys, kfid[i], pose7, info, CAP_KP, kid.data(), kpx.data(), k3.data());
            const int32_t nc = ref_system_covisibility(sys, kfid[i], 256, cov.data());
            need(n >= 0 && n <= CAP_KP && nc >= 0 && nc <= 256, "a keyframe table outgrew its capacity");
            fwrite(&kfid[i], 4, 1, o); fwrite(pose7, 8, 7, o); fwrite(info, 4, 6, o);
            fwrite(&n, 4, 1, o); fwrite(kid.data(), 4, n, o); fwrite(kpx.data(), 4, 2 * (size_t) n, o); fwrite(k3.data(), 1, n, o);
            fwrite(&nc, 4, 1, o); fwrite(cov.data(), 4, 2 * (size_t) nc, o);
        }
    }
    ref_system_destroy(sys);
    need(fclose(o) == 0, "writing the records failed");
    return 0;
}
