// nav_host.hip — the step immediately AFTER the tracking path: bit synchronisation, 20 ms nav-bit
// accumulation on the prompt I and preamble correlation of src/decoding.rs:8,40-227 (a legacy file outside the
// reference's module tree), plus the GPS word parity of :259-352.  Tiny integer work on one f32 per channel-epoch:
// it stays on the host (no kernel), fed by gm_trk_update_all's prompt outputs.
#include <deque>
#include <vector>

#include "gm_internal.h"

namespace {
constexpr int BIT_MS = 20;                       // GPS_L1_CA_BIT_PERIOD_MS (gps_property_constants.rs:9)
constexpr int PREAMBLE_BITS = 8;                 // GPS_CA_PREAMBLE_LENGTH_BITS (:15)
constexpr int8_t PREAMBLE[8] = {1, -1, -1, -1, 1, -1, 1, 1};   // GPS_CA_PREAMBLE (:13)
constexpr uint64_t BIT_SYNC_THRESHOLD = 30;      // decoding.rs:8
constexpr uint64_t LOOP_MS = 10;                 // tracking::LOOP_MS (tracking_bk.rs:21)
}  // namespace

struct gm_nav_sync {                             // NavSyncStatus (decoding.rs:40-63)
    int mode = GM_NAV_FAITHFUL;
    bool flag_bit_sync = false, flag_frame_sync = false, sync_sw = false, loop_sw = false;
    uint64_t biti = 0, frame_sync_ind = 0, bit_code_cnt = 0, sf_buffer_loc = 0, sf_cnt = 0, sf_start_biti = 0,
             tow_expected_ind = 0;
    uint64_t histogram[BIT_MS] = {};             // bit_sync_buff
    float i_p = 0.0f;
    int8_t polarity = -1;
    std::vector<int8_t> frame_bits;
    std::vector<uint64_t> buffer_loc_biti;
    std::deque<int8_t> buff_preamble;
};

extern "C" {

int gm_nav_sync_create(int mode, gm_nav_sync** out) {   // NavSyncStatus::new :68-100
    if (!out || (mode != GM_NAV_FAITHFUL && mode != GM_NAV_FIXED)) return GM_ERR_INVALID_ARG;
    *out = new gm_nav_sync;
    (*out)->mode = mode;
    return GM_OK;
}

int gm_nav_sync_destroy(gm_nav_sync* s) {
    delete s;
    return GM_OK;
}

int gm_nav_sync_update(gm_nav_sync* s, float old_i_prompt, float i_prompt, uint64_t cnt, uint64_t buff_loc,
                       gm_nav_status* out) {   // nav_decoding :102-145
    if (!s) return GM_ERR_INVALID_ARG;
    s->biti = cnt % BIT_MS;                                                            // :114
    if (!s->flag_bit_sync && cnt > uint64_t(1.0f / 1.0e-3f)) {                         // :115-118
        bool synced = false;                                                           // check_bit_sync :164-182
        if (old_i_prompt * i_prompt < 0.0f) {
            s->histogram[s->biti] += 1;
            int arg = BIT_MS - 1;                      // Iterator::max_by returns the LAST maximum: scan downwards, strict >
            for (int i = BIT_MS - 2; i >= 0; --i)
                if (s->histogram[i] > s->histogram[arg]) arg = i;
            s->frame_sync_ind = uint64_t(arg);
            synced = s->histogram[arg] == BIT_SYNC_THRESHOLD;
        }
        s->flag_bit_sync = synced;
    }
    if (s->flag_bit_sync) {                                                            // bit_accumulation :184-214
        s->sync_sw = false;
        if (s->biti == s->frame_sync_ind) { s->bit_code_cnt = 1; s->i_p = i_prompt; }
        else s->i_p += i_prompt;
        s->loop_sw = s->bit_code_cnt % LOOP_MS == 0;
        const uint64_t boundary = s->frame_sync_ind + BIT_MS - 1;                      // :203-205: no modulo in the reference
        if (s->biti == (s->mode == GM_NAV_FIXED ? boundary % BIT_MS : boundary)) {
            const int8_t bit = s->i_p > 0.0f ? 1 : -1;
            s->frame_bits.push_back(bit);
            s->buffer_loc_biti.push_back(buff_loc);
            s->sync_sw = true;
            if (!s->flag_frame_sync) {
                s->buff_preamble.push_back(bit);                                       // grows without bound in the reference
                if (s->mode == GM_NAV_FIXED && s->buff_preamble.size() > size_t(PREAMBLE_BITS)) s->buff_preamble.pop_front();
            }
        }
        s->bit_code_cnt += 1;
    }
    if (s->sync_sw) {                                                                  // :129-145
        if (!s->flag_frame_sync && s->buff_preamble.size() == size_t(PREAMBLE_BITS)) {
            int corr = 0;                                                              // check_preamble_syn :216-227
            for (int x = 0; x < PREAMBLE_BITS; ++x) corr += s->buff_preamble[x] * PREAMBLE[x % 8];
            if (corr == PREAMBLE_BITS || corr == -PREAMBLE_BITS) {
                s->polarity = int8_t(corr > 0 ? 1 : -1);
                s->flag_frame_sync = true;
            }
        }
        if (s->flag_frame_sync) {
            s->sf_buffer_loc = buff_loc;
            s->sf_cnt = cnt;
            s->sf_start_biti = s->frame_bits.size() - PREAMBLE_BITS;
            s->tow_expected_ind = cnt + 30 * 20;       // GPS_WORD_BITS * GPS_CA_TELEMETRY_SYMBOLS_PER_BIT
        }
    }
    if (out) {
        out->flag_bit_sync = s->flag_bit_sync; out->flag_frame_sync = s->flag_frame_sync; out->sync_sw = s->sync_sw;
        out->bit = s->sync_sw ? s->frame_bits.back() : int8_t(0);
        out->polarity = s->polarity;
        out->frame_sync_ind = uint32_t(s->frame_sync_ind);
        out->n_frame_bits = uint64_t(s->frame_bits.size());
        out->i_p = s->i_p;
        out->sf_cnt = s->sf_cnt; out->sf_start_biti = s->sf_start_biti; out->tow_expected_ind = s->tow_expected_ind;
    }
    return GM_OK;
}

int gm_nav_sync_update_many(gm_nav_sync* s, float old_i_prompt0, const float* i_prompt, size_t stride, size_t n, uint64_t cnt0,
                            uint64_t buff_loc, gm_nav_status* out, int64_t* first_bit_sync, int64_t* first_frame_sync) {
    if (!s || (!i_prompt && n) || !stride) return GM_ERR_INVALID_ARG;
    if (first_bit_sync) *first_bit_sync = -1;
    if (first_frame_sync) *first_frame_sync = -1;
    float old = old_i_prompt0;
    for (size_t k = 0; k < n; ++k) {            // nav_decoding once per epoch (:102-145), as the caller's own loop would
        const bool had_bit = s->flag_bit_sync, had_frame = s->flag_frame_sync;
        const float cur = i_prompt[k * stride];
        if (int rc = gm_nav_sync_update(s, old, cur, cnt0 + k, buff_loc, k + 1 == n ? out : nullptr)) return rc;
        if (first_bit_sync && !had_bit && s->flag_bit_sync && *first_bit_sync < 0) *first_bit_sync = int64_t(k);
        if (first_frame_sync && !had_frame && s->flag_frame_sync && *first_frame_sync < 0) *first_frame_sync = int64_t(k);
        old = cur;
    }
    return GM_OK;
}

int gm_nav_sync_frame_bits(gm_nav_sync* s, int8_t* bits, size_t cap, size_t* n) {
    if (!s || !n) return GM_ERR_INVALID_ARG;
    *n = s->frame_bits.size();
    if (bits)
        for (size_t i = 0; i < s->frame_bits.size() && i < cap; ++i) bits[i] = s->frame_bits[i];
    return GM_OK;
}

int gm_nav_sync_histogram(gm_nav_sync* s, uint64_t hist[20]) {
    if (!s || !hist) return GM_ERR_INVALID_ARG;
    for (int i = 0; i < BIT_MS; ++i) hist[i] = s->histogram[i];
    return GM_OK;
}

// parity_check :259-352 on 32 symbols in +-1 form: [D29*, D30*, d1..d24, D25..D30].  The six products are the
// IS-GPS-200 equations in the reference's multiplicative form.  *ok: every product equals its parity symbol;
// *ref_sum_zero: the reference's own test (:348-350, the i8 SUM of the six differences is zero — +2 and -2 cancel).
int gm_nav_parity_check(const int8_t bits[32], int* ok, int* ref_sum_zero) {
    if (!bits || !ok) return GM_ERR_INVALID_ARG;
    // participating symbols of the six products, as index lists terminated by -1 (decoding.rs:262-346)
    static const int8_t terms[6][17] = {
        {0, 2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24, -1},
        {1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22, 25, -1},
        {0, 2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23, -1},
        {1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24, -1},
        {1, 2, 4, 6, 7, 8, 10, 11, 15, 16, 17, 18, 19, 22, 23, 25, -1},
        {0, 4, 6, 7, 9, 10, 11, 12, 14, 16, 20, 23, 24, 25, -1}};
    uint32_t neg = 0;                                   // bit i set <-> bits[i] == -1
    for (int i = 0; i < 32; ++i) {
        if (bits[i] != 1 && bits[i] != -1) return GM_ERR_INVALID_ARG;
        if (bits[i] < 0) neg |= 1u << i;
    }
    bool all = true;
    int sum = 0;
    for (int k = 0; k < 6; ++k) {
        uint32_t mask = 0;
        for (int j = 0; terms[k][j] >= 0; ++j) mask |= 1u << terms[k][j];
        const int prod = (__builtin_popcount(neg & mask) & 1) ? -1 : 1;   // product of +-1 = parity of the minus signs
        if (prod != bits[26 + k]) all = false;
        sum += prod - bits[26 + k];
    }
    *ok = all ? 1 : 0;
    if (ref_sum_zero) *ref_sum_zero = int8_t(sum) == 0 ? 1 : 0;
    return GM_OK;
}

}  // extern "C"
