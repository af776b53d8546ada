// bhw.hpp -- C++ host mirror of the reference's operator interface, header-only, on top of the C ABI (bhw.h).
//
// The reference is compiled code (VHDL entities + C++ bit-models), so the host side above the ABI is C++:
//   bhw::win_selector   <->  entity win_selector            src/win_selector.vhd:60-87
//   bhw::win_function() <->  HLS top win_function()         hls/windows/win_function.h:65-69
//   bhw::cordic()       <->  cordic()                       cpp/cordic_sincos.cpp:10, hls/cordic/cordic.cpp:45;
//                            entities cordic_dds / cordic_dds48 / cordic_dds_scaled (model BHW_MODEL_VHDL / _DDS48 / _SCALED)
//   bhw::cordic_atan2() <->  entity cordic_atan2            src/cordic_atan2.vhd:64-76
// Same names, argument meaning and error behaviour (unknown win_type -> zeros, like win_empty,
// hls/windows/win_function.cpp:159-165,417-419).  All arithmetic runs in the HIP kernels behind the ABI.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "bhw.h"

namespace bhw {

struct error : std::runtime_error {
    int code;
    error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
    if (rc != BHW_OK) throw error(rc, std::string(bhw_strerror(rc)) + ": " + bhw_last_error());
}

// entity win_selector: generics become constructor arguments under the reference's names.
class win_selector {
public:
    win_selector(unsigned PHI_WIDTH, unsigned DAT_WIDTH, const std::string &WIN_TYPE, const std::string &SIN_TYPE = "CORDIC",
                 unsigned LUT_SIZE = 9, const std::string &XSERIES = "ULTRA", int device = 0)
        : device_(device)
    {
        uint32_t wt = 0;
        if (WIN_TYPE == "HAMMING") wt = BHW_WIN_HAMMING;
        else if (WIN_TYPE == "HANN") wt = BHW_WIN_HANN;
        else if (WIN_TYPE == "BH3TERM") wt = BHW_WIN_BH3;
        else if (WIN_TYPE == "BH4TERM") wt = BHW_WIN_BH4;
        else if (WIN_TYPE == "BH5TERM") wt = BHW_WIN_BH5;
        else if (WIN_TYPE == "BH7TERM") wt = BHW_WIN_BH7;
        else throw error(BHW_ERR_BADARG, "WIN_TYPE " + WIN_TYPE);
        if (XSERIES != "ULTRA" && XSERIES != "7SERIES") throw error(BHW_ERR_BADARG, "XSERIES " + XSERIES);
        int rc = bhw_params_init(&p_, wt, PHI_WIDTH, DAT_WIDTH);
        if (rc != BHW_OK && p_.n_terms == 0) check(rc);
        // SIN_TYPE reaches only hamming_win and bh_win_3term (src/win_selector.vhd:93-135); the 4/5/7-term entities have
        // no such generic (:137-199), so "TAYLOR" there is still the CORDIC design.  "TAYLOR_ALL": extension, see bhw.h.
        if (SIN_TYPE == "TAYLOR_ALL") p_.sin_type = BHW_SIN_TAYLOR_ALL;
        else if (SIN_TYPE == "TAYLOR") p_.sin_type = (p_.n_terms <= 3) ? BHW_SIN_TAYLOR : BHW_SIN_CORDIC;
        else if (SIN_TYPE != "CORDIC") throw error(BHW_ERR_BADARG, "SIN_TYPE " + SIN_TYPE);
        p_.lut_size = LUT_SIZE;
        check(bhw_params_validate(&p_));
    }

    // the AA0..AA6 ports (integer, caller-scaled)
    void set_AA(const std::vector<int32_t> &aa)
    {
        for (size_t k = 0; k < 7; ++k) p_.aa[k] = k < aa.size() ? aa[k] : 0;
    }
    void set_model(uint32_t model, uint32_t combine, uint32_t precision = 1)
    {
        p_.model = model;
        p_.combine = combine;
        p_.precision = precision;
        check(bhw_params_validate(&p_));
    }
    void RESET() { phase_ = 0; }
    uint64_t length() const { return 1ull << p_.phi_width; }

    // ENABLE high for `count` clocks: the next `count` values of DT_WIN into device memory (async on `stream`).
    void ENABLE(uint64_t count, int32_t *d_out, void *stream = nullptr)
    {
        check(bhw_generate_device(&p_, device_, stream, phase_, count, d_out));
        phase_ = (phase_ + count) % length();
    }
    // same, delivered to host memory
    std::vector<int32_t> ENABLE(uint64_t count)
    {
        std::vector<int32_t> v(count);
        check(bhw_generate_to_host(&p_, device_, phase_, count, v.data()));
        phase_ = (phase_ + count) % length();
        return v;
    }
    // the multiplier stage behind DT_WIN: d_y[i] = (d_x[i] * DT_WIN) >> shift for the next `count` clocks
    void APPLY(uint64_t count, const int32_t *d_x, int32_t *d_y, unsigned shift, void *stream = nullptr)
    {
        check(bhw_apply_device(&p_, device_, stream, phase_, count, d_x, d_y, shift));
        phase_ = (phase_ + count) % length();
    }
    // Every lazy step of later calls done now (ROM upload, scratch, packed-table verification): bhw_prepare_device.
    void PREPARE(void *stream = nullptr) { check(bhw_prepare_device(&p_, device_, stream)); }
    // One window over several devices without a collective: what part `part` of `n_parts` owns ...
    std::vector<bhw_segment> SEGMENTS(uint32_t part, uint32_t n_parts) const
    {
        uint32_t n = 0;
        check(bhw_part_segments(&p_, part, n_parts, nullptr, 0, &n));
        std::vector<bhw_segment> v(n);
        check(bhw_part_segments(&p_, part, n_parts, v.data(), n, &n));
        return v;
    }
    // ... and those coefficients written into d_window, the base of a full-length (2^PHI_WIDTH) device buffer
    void ENABLE_PART(uint32_t part, uint32_t n_parts, int32_t *d_window, void *stream = nullptr)
    {
        check(bhw_generate_part_device(&p_, device_, stream, part, n_parts, d_window, nullptr));
    }
    // the whole window on this selector's device from its n_parts parts (d_windows[g] on src_devices[g]): peer copies of the
    // owned segments, no collective (bhw_gather_parts_device)
    void GATHER_PARTS(uint32_t n_parts, const int *src_devices, const int32_t *const *d_windows, int32_t *d_dst, void *stream = nullptr)
    {
        check(bhw_gather_parts_device(&p_, n_parts, src_devices, d_windows, device_, stream, d_dst));
    }
    // AA0..AA6 from one of the named coefficient sets of the reference's comments / README (bhw_coeffs_preset); the window
    // type must be the preset's
    void AA_PRESET(uint32_t preset)
    {
        uint32_t wt = 0;
        int32_t aa[7];
        check(bhw_coeffs_preset(preset, p_.dat_width, &wt, nullptr, aa));
        if (wt != p_.win_type) throw error(BHW_ERR_BADARG, "preset belongs to another window type");
        for (int k = 0; k < 7; ++k) p_.aa[k] = aa[k];
    }
    const bhw_params &params() const { return p_; }

private:
    bhw_params p_{};
    uint64_t phase_ = 0;
    int device_;
};

// HLS top swept over i = i0 .. i0+count-1.
inline std::vector<int32_t> win_function(char win_type, uint64_t i0, uint64_t count, unsigned NPHASE, unsigned NWIDTH, int device = 0)
{
    std::vector<int32_t> v(count, 0);
    bhw_params p;
    if (bhw_params_init(&p, (uint32_t)(unsigned char)win_type, NPHASE, NWIDTH) != BHW_OK) {
        if (p.n_terms == 0) return v;  // win_empty
        check(bhw_params_validate(&p));
    }
    check(bhw_generate_to_host(&p, device, i0, count, v.data()));
    return v;
}

// cordic() swept over theta: returns {sin, cos} vectors.  model = BHW_MODEL_CPP reproduces cpp/cordic_sincos.cpp.
inline void cordic(uint32_t model, unsigned PHASE_WIDTH, unsigned DATA_WIDTH, uint64_t theta0, uint64_t count,
                   std::vector<int32_t> &s, std::vector<int32_t> &c, int device = 0)
{
    bhw_params p;
    bhw_params_init(&p, BHW_WIN_HAMMING, PHASE_WIDTH, DATA_WIDTH);
    p.model = model;
    s.resize(count);
    c.resize(count);
    check(bhw_sincos_to_host(&p, device, theta0, count, s.data(), c.data()));
}

// entity cordic_atan2 (src/cordic_atan2.vhd:64-76) over host vectors VEC_DX, VEC_DY -> PHI_DT.
inline std::vector<int32_t> cordic_atan2(unsigned PRECISION, unsigned INPUT_WIDTH, unsigned ANGLE_WIDTH,
                                         const std::vector<int32_t> &VEC_DX, const std::vector<int32_t> &VEC_DY, int device = 0)
{
    if (VEC_DX.size() != VEC_DY.size()) throw error(BHW_ERR_BADARG, "VEC_DX / VEC_DY lengths differ");
    bhw_atan2_params p{(uint32_t)sizeof(bhw_atan2_params), PRECISION, INPUT_WIDTH, ANGLE_WIDTH};
    std::vector<int32_t> phi(VEC_DX.size());
    check(bhw_atan2_to_host(&p, device, VEC_DX.size(), VEC_DX.data(), VEC_DY.data(), phi.data()));
    return phi;
}

} // namespace bhw
