// Differential test of same::combine (same_transport.cpp: word-wide bit voting, specialised stretches) against the plain byte-by-byte
// walk of estimate_message + combine (rx/combiner.rs:32-80, 154-203) on random bursts: bit errors, high bits, garbage bytes, ragged
// lengths.  Built with -fsanitize=address,undefined by tests/test_host_sanitizers.py.   ./combine_fuzz [cases]
#include <cstdlib>
#include "../../sameold_amd/csrc/same_transport.h"
#include <cstdio>
#include <random>
#include <algorithm>
using namespace same;
// reference: the byte-by-byte walk of estimate_message + combine (rx/combiner.rs:32-80, 154-203), no fast paths
static void parse_ref(const uint8_t *b, size_t n, const uint8_t *errs, const uint8_t *counts, MessageResult *out)
{
    out->kind = 0; out->err = 0; out->len = 0; out->offset_time = 0; out->parity_errors = 0; out->voting_bytes = 0;
    for (size_t i = 0; i < n; ++i) if (b[i] & 0x80) { out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 1; return; }
    if (n >= 5 && std::memcmp(b, "ZCZC-", 5) == 0) {
        size_t off = 0, hl = 0;
        if (!check_header(b, n, &off, &hl)) { out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 3; return; }
        out->kind = SAME_TRANSPORT_MSG_START; out->len = (uint32_t)hl; std::memcpy(out->text, b, hl); out->offset_time = (uint32_t)off;
        for (size_t i = 0; i < hl; ++i) { out->parity_errors += errs[i]; out->voting_bytes += counts[i] >= 3 ? 1u : 0u; }
    } else if (n >= 2 && b[0] == 'N' && b[1] == 'N') out->kind = SAME_TRANSPORT_MSG_END;
    else { out->kind = SAME_TRANSPORT_MSG_ERR; out->err = 2; }
}
static bool combine_ref(const BurstBuf *bursts, uint32_t nbursts, MessageResult *res)
{
    uint8_t msg[kMaxMessageLength], cnt[kMaxMessageLength], errs[kMaxMessageLength];
    const size_t nb = std::min<size_t>(nbursts, 3);
    size_t n = 0;
    while (n < kMaxMessageLength) {
        uint8_t cur[3]; uint32_t k = 0; bool msb = false;
        for (size_t i = 0; i < nb; ++i) if (n < bursts[i].len) cur[k++] = bursts[i].data[n];
        for (uint32_t i = 0; i < k; ++i) { msb |= (cur[i] & 0x80) != 0; cur[i] &= 0x7f; }
        if (k == 0) break;
        uint8_t est; uint32_t be = 0;
        if (k == 1) est = cur[0]; else if (k == 2) bit_vote_detect(cur[0], cur[1], &est, &be); else bit_vote_correct(cur[0], cur[1], cur[2], &est, &be);
        if (!is_allowed_byte(est)) break;
        msg[n] = est; cnt[n] = (uint8_t)k; errs[n] = (uint8_t)(be + (msb ? 1u : 0u)); ++n;
    }
    if (n == 0) return false;
    size_t good = 0; while (good < n && cnt[good] >= 2) ++good;
    parse_ref(msg, good, errs, cnt, res);
    if (res->kind != SAME_TRANSPORT_MSG_ERR) return true;
    if (n >= 2 && msg[0] == 'N' && msg[1] == 'N') { res->kind = SAME_TRANSPORT_MSG_END; res->err = 0; res->len = 0; res->offset_time = 0; res->parity_errors = 0; res->voting_bytes = 0; return true; }
    return good != 0;
}
int main(int argc, char **argv)
{
    std::mt19937_64 rng(12345);
    const char *hdrs[] = {"ZCZC-PEP-YED-355586-074017-633245-699575-585044+3110-8478384-X2//TA8B-", "ZCZC-WXR-RWT-012345+0030-1231200-KABC/NWS-", "NNNN", "ZCZC-EAS-DMO-999000-123456+0015-0011122-WXYZ    -"};
    size_t n_cases = 0, n_msg = 0;
    const int n_it = argc > 1 ? std::atoi(argv[1]) : 200000;
    for (int it = 0; it < n_it; ++it) {
        BurstBuf b[3]; const uint32_t nb = 1 + rng() % 3;
        const char *h = hdrs[rng() % 4]; const size_t hl = strlen(h);
        for (uint32_t i = 0; i < nb; ++i) {
            size_t len = hl;
            const unsigned mode = rng() % 8;
            if (mode == 0) len = rng() % (hl + 1); else if (mode == 1) len = std::min<size_t>(hl + rng() % 40, kMaxMessageLength);
            b[i].len = (uint32_t)len;
            for (size_t k = 0; k < len; ++k) b[i].data[k] = k < hl ? (uint8_t)h[k] : (uint8_t)('A' + rng() % 26);
            const unsigned flips = rng() % 6 == 0 ? rng() % 12 : rng() % 3;          // bit errors, high bits included
            for (unsigned f = 0; f < flips && len; ++f) b[i].data[rng() % len] ^= (uint8_t)(1u << (rng() % 8));
            if (rng() % 16 == 0 && len) b[i].data[rng() % len] = (uint8_t)(rng() & 0xff);   // a garbage byte
        }
        MessageResult r0, r1; std::memset(&r0, 0, sizeof r0); std::memset(&r1, 0, sizeof r1);
        const bool k0 = combine_ref(b, nb, &r0), k1 = combine(b, nb, &r1);
        ++n_cases;
        if (k0 != k1 || (k0 && !(r0 == r1))) { printf("MISMATCH at case %d: %d %d kinds %u %u len %u %u perr %u %u vb %u %u\n", it, k0, k1, r0.kind, r1.kind, r0.len, r1.len, r0.parity_errors, r1.parity_errors, r0.voting_bytes, r1.voting_bytes); return 1; }
        n_msg += k0 && r0.kind == SAME_TRANSPORT_MSG_START;
    }
    printf("%zu cases equal (%zu headers decoded)\n", n_cases, n_msg);
}
