// resample_table.hpp -- host-side construction of the per-axis contribution tables that the
// device resampler consumes.
//
// Behavioural contract: FRawScaleWeightsTable's constructor in the reference
// (src/frawscale.cpp:8-112) with the five filter kernels of src/frawscale.h:60-172.  The table
// is tiny (dst_len x <=window doubles), depends only on (filter, dst_len, src_len), and its
// doubles must be the same doubles the reference computes, so it is built on the host in plain
// IEEE double arithmetic (this TU is compiled with -ffp-contract=off) and cached/uploaded by
// the C-ABI layer.  Device layout: for output coordinate u,
//     first[u]           first source index
//     taps[u]            number of taps actually summed (after trailing-zero trimming)
//     weight[u*stride+t] weight of source index first[u]+t,  t < taps[u]
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace srcnn {

enum FilterKind { kNearest = 0, kBilinear = 1, kBicubic = 2, kLanczos3 = 3, kBspline = 4 };

class ResampleKernel {
public:
    explicit ResampleKernel(int kind) : kind_(kind)
    {
        switch (kind) {
        case kNearest:  support_ = 0.5; break;            // frawscale.h:64
        case kBilinear: support_ = 1.0; break;            // frawscale.h:75
        case kLanczos3: support_ = 3.0; break;            // frawscale.h:127
        case kBspline:  support_ = 2.0; break;            // frawscale.h:157
        default: {                                        // Mitchell-Netravali B=C=1/3, frawscale.h:95-107
            kind_ = kBicubic;
            support_ = 2.0;
            const double B = 1 / (double)3, Cc = 1 / (double)3;
            in_[0] = (6 - 2 * B) / 6;
            in_[1] = (-18 + 12 * B + 6 * Cc) / 6;
            in_[2] = (12 - 9 * B - 6 * Cc) / 6;
            out_[0] = (8 * B + 24 * Cc) / 6;
            out_[1] = (-12 * B - 48 * Cc) / 6;
            out_[2] = (6 * B + 30 * Cc) / 6;
            out_[3] = (-B - 6 * Cc) / 6;
        }
        }
    }
    double support() const { return support_; }
    double operator()(double x) const
    {
        x = std::fabs(x);
        switch (kind_) {
        case kNearest:  return x <= support_ ? 1.0 : 0.0;
        case kBilinear: return x < support_ ? support_ - x : 0.0;
        case kLanczos3: return x < support_ ? sinc(x) * sinc(x / support_) : 0.0;
        case kBspline:
            if (x < 1) return (4 + x * x * (-6 + 3 * x)) / 6;
            if (x < 2) { const double t = 2 - x; return t * t * t / 6; }
            return 0;
        default:
            if (x < 1) return in_[0] + x * x * (in_[1] + x * in_[2]);
            if (x < 2) return out_[0] + x * (out_[1] + x * (out_[2] + x * out_[3]));
            return 0;
        }
    }
private:
    static double sinc(double v)
    {
        if (v != 0) { v *= 3.1415926535897932384626433832795; return std::sin(v) / v; }
        return 1;
    }
    int kind_;
    double support_;
    double in_[3] = {0, 0, 0}, out_[4] = {0, 0, 0, 0};
};

struct AxisTable {
    unsigned dst_len = 0, src_len = 0;
    int filter = kBicubic;
    int window = 0;              // reference's _WindowSize
    int stride = 0;              // doubles per row (window + 1, like the reference's allocation)
    std::vector<int32_t> first;  // Left
    std::vector<int32_t> last;   // Right (inclusive, after trimming)
    std::vector<int32_t> taps;   // last - first + 1
    std::vector<double> weight;  // dst_len * stride
    int max_taps = 0;
};

inline AxisTable build_axis_table(int filter, unsigned dst_len, unsigned src_len)
{
    AxisTable t;
    t.dst_len = dst_len; t.src_len = src_len; t.filter = filter;
    const ResampleKernel kern(filter);
    const double ratio = double(dst_len) / double(src_len);
    double reach = kern.support(), squeeze = 1.0;
    if (ratio < 1.0) { reach = kern.support() / ratio; squeeze = ratio; }   // frawscale.cpp:25-33
    t.window = 2 * (int)std::ceil(reach) + 1;                               // frawscale.cpp:35
    t.stride = t.window + 1;
    t.first.resize(dst_len); t.last.resize(dst_len); t.taps.resize(dst_len);
    t.weight.assign(size_t(dst_len) * t.stride, 0.0);
    const double phase0 = (0.5 / ratio) - 0.5;                              // frawscale.cpp:46

    for (unsigned u = 0; u < dst_len; ++u) {
        const double mid = (double)u / ratio + phase0;
        int a = std::max(0, (int)std::floor(mid - reach));
        int b = std::min((int)std::ceil(mid + reach), int(src_len) - 1);
        if (b - a + 1 > t.window) {
            // the reference compares against "int(uSrcSize) - 1 / 2" == uSrcSize (integer 1/2 is 0),
            // so the left edge always moves (frawscale.cpp:55-65)
            if (a < int(src_len) - 1 / 2) ++a; else --b;
        }
        double* row = &t.weight[size_t(u) * t.stride];
        double norm = 0;
        for (int s = a; s <= b; ++s) {
            const double wv = squeeze * kern(squeeze * (mid - (double)s));
            row[s - a] = wv;
            norm += wv;
        }
        int bb = b;
        if (norm > 0 && norm != 1) {
            for (int s = a; s <= b; ++s) row[s - a] /= norm;
            int k = b - a;
            while (row[k] == 0) {           // only TRAILING zero taps are dropped (frawscale.cpp:95-107)
                --bb; --k;
                if (bb == a) break;
            }
        }
        t.first[u] = a; t.last[u] = bb; t.taps[u] = bb - a + 1;
        // taps beyond the window read as weight 0.0 in the reference's getWeight(); they cannot occur
        // for the ranges built above (at most window+1 candidates, one removed), but clamp anyway.
        if (t.taps[u] > t.window) t.taps[u] = t.window;
        t.max_taps = std::max(t.max_taps, (int)t.taps[u]);
    }
    return t;
}

}  // namespace srcnn
