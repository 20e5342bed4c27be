// Host-side weight repacking: PyTorch layouts -> the layouts the kernels read.
#include <cmath>
#include <cstring>

#include "common.h"

namespace gdpso {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
const char *last_error_cstr() { return g_error.c_str(); }

DeviceArena::~DeviceArena() {
    for (void *p : blocks) (void)hipFree(p);
}

int DeviceArena::upload(const void *host, size_t bytes, const void **dev) {
    void *p = nullptr;
    GDPSO_HIP(hipMalloc(&p, bytes ? bytes : 4));
    blocks.push_back(p);
    if (bytes) GDPSO_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    *dev = p;
    return GDPSO_OK;
}

// BatchNorm2d in eval mode: y = (x - mean) / sqrt(var + 1e-5) * gamma + beta, with an
// optional bias added to x first (ConvTranspose2d layers of the generator have one).
void fold_bn(int c, const float *bias, const float *gamma, const float *beta, const float *mean,
             const float *var, std::vector<float> &scale, std::vector<float> &shift) {
    scale.resize(c);
    shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const double a = (double)gamma[i] / std::sqrt((double)var[i] + 1e-5);
        const double b = bias ? (double)bias[i] : 0.0;
        scale[i] = (float)a;
        shift[i] = (float)((b - (double)mean[i]) * a + (double)beta[i]);
    }
}

static bool tap_ever_in_range(int k, int stride, int pad, int n_out, int n_in) {
    for (int o = 0; o < n_out; ++o) {
        const int i = o * stride - pad + k;
        if (i >= 0 && i < n_in) return true;
    }
    return false;
}

void make_conv(ConvLayer &L, const float *w, int cout, int cin, int kh, int kw, int stride, int pad,
               int hin, int win, int act, std::vector<float> &packed, int cin_pad, int cout_pad) {
    if (cin_pad < cin) cin_pad = cin;
    if (cout_pad < cout) cout_pad = cout;
    std::memset(&L, 0, sizeof(L));
    L.cin = cin_pad;
    L.cout = cout_pad;
    L.hin = hin;
    L.win = win;
    L.hout = L.hm = (hin + 2 * pad - kh) / stride + 1;
    L.wout = L.wm = (win + 2 * pad - kw) / stride + 1;
    L.in_stride = stride;
    L.out_mul = 1;
    L.nphase = 1;
    L.act = act;
    L.flops_nominal = 2.0 * L.hout * L.wout * cout * cin * kh * kw;
    ConvPhase &P = L.phase[0];
    std::vector<int> live;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx)
            if (tap_ever_in_range(ky, stride, pad, L.hm, hin) && tap_ever_in_range(kx, stride, pad, L.wm, win)) {
                P.dy[live.size()] = (int8_t)(ky - pad);
                P.dx[live.size()] = (int8_t)(kx - pad);
                live.push_back(ky * kw + kx);
            }
    P.ntaps = (int)live.size();
    const size_t ktot = (size_t)P.ntaps * cin_pad;
    packed.assign((size_t)cout_pad * ktot, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int t = 0; t < P.ntaps; ++t)
            for (int ci = 0; ci < cin; ++ci)
                packed[(size_t)co * ktot + (size_t)t * cin_pad + ci] = w[((size_t)co * cin + ci) * kh * kw + live[t]];
}

// ConvTranspose2d(k=4, s=2, p=1): oy = 2*iy - 1 + ky.  Output row 2m takes (iy, ky) in
// {(m,1), (m-1,3)}; row 2m+1 takes {(m,2), (m+1,0)}; likewise along x.  Each of the four
// output phases is therefore a 2x2-tap convolution over the input.
void make_convt_k4s2p1(ConvLayer &L, const float *w, int cin, int cout, int hin, int win, int act,
                       std::vector<float> &packed) {
    std::memset(&L, 0, sizeof(L));
    L.cin = cin;
    L.cout = cout;
    L.hin = L.hm = hin;
    L.win = L.wm = win;
    L.hout = 2 * hin;
    L.wout = 2 * win;
    L.in_stride = 1;
    L.out_mul = 2;
    L.nphase = 4;
    L.act = act;
    L.flops_nominal = 2.0 * L.hout * L.wout * cout * cin * 4;
    static const int d_of[2][2] = {{0, -1}, {0, 1}};
    static const int k_of[2][2] = {{1, 3}, {2, 0}};
    const size_t ktot = (size_t)4 * cin;
    packed.assign((size_t)4 * cout * ktot, 0.f);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int ph = py * 2 + px;
            ConvPhase &P = L.phase[ph];
            P.ntaps = 4;
            P.off_y = py;
            P.off_x = px;
            P.wt_offset = (long)ph * cout * ktot;
            for (int ty = 0; ty < 2; ++ty)
                for (int tx = 0; tx < 2; ++tx) {
                    const int t = ty * 2 + tx;
                    P.dy[t] = (int8_t)d_of[py][ty];
                    P.dx[t] = (int8_t)d_of[px][tx];
                    const int ky = k_of[py][ty], kx = k_of[px][tx];
                    for (int co = 0; co < cout; ++co)
                        for (int ci = 0; ci < cin; ++ci)
                            packed[P.wt_offset + (size_t)co * ktot + (size_t)t * cin + ci] =
                                w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
                }
        }
}

int upload_conv(DeviceArena &arena, ConvLayer &L, const std::vector<float> &packed,
                const std::vector<float> &scale, const std::vector<float> &shift) {
    std::vector<int> table((size_t)L.nphase * (4 + MAX_TAPS), 0);
    for (int p = 0; p < L.nphase; ++p) {
        int *t = table.data() + (size_t)p * (4 + MAX_TAPS);
        t[0] = L.phase[p].ntaps;
        t[1] = L.phase[p].off_y;
        t[2] = L.phase[p].off_x;
        t[3] = (int)L.phase[p].wt_offset;
        for (int k = 0; k < L.phase[p].ntaps; ++k)
            t[4 + k] = ((int)L.phase[p].dy[k] & 0xffff) | ((int)L.phase[p].dx[k] << 16);
    }
    int rc;
    if ((rc = arena.upload(packed, &L.wt))) return rc;
    if ((rc = arena.upload(scale, &L.scale))) return rc;
    if ((rc = arena.upload(shift, &L.shift))) return rc;
    if ((rc = arena.upload(table.data(), table.size() * sizeof(int), (const void **)&L.table))) return rc;
    return GDPSO_OK;
}

}  // namespace gdpso

extern "C" {
int gdpso_abi_version(void) { return GDPSO_ABI_VERSION; }
const char *gdpso_last_error(void) { return gdpso::last_error_cstr(); }
size_t gdpso_struct_size(int which) { return which == 0 ? sizeof(gdpso_swarm) : which == 1 ? sizeof(gdpso_run) : 0; }
}
