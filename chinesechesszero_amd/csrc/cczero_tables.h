// cczero_tables.h -- compile-time action-space tables for the gfx950 engine.
//
// The 2086-move action space follows reference tools.py:172-272 (get_all_legal_moves): ids
// 0..2037 enumerate, for every from-square in rank-major order, same-file destinations by rank,
// same-rank destinations by file, then the eight knight jumps in the reference's listed order
// (off-board ones dropped); 2038..2053 are the advisor moves and 2054..2085 the bishop moves in
// the reference's listed order. square = file + 9*rank (tools.py:91).
// The file mirror (tools.py:133-166 flip, collect.py:118-123 flip map) is derived from it.
#pragma once
#include <stdint.h>

namespace ccz {

constexpr int kNMoves = 2086;
constexpr int kNSq = 90;

struct Tables {
    uint8_t from[kNMoves];
    uint8_t to[kNMoves];
    uint16_t inv[kNSq * kNSq]; // (from*90+to) -> id, 0xFFFF if the pair is not an action
    uint16_t flip[kNMoves];    // id -> id of the file-mirrored move
};

constexpr Tables make_tables()
{
    Tables t{};
    for (int i = 0; i < kNSq * kNSq; ++i) t.inv[i] = 0xFFFF;
    int n = 0;
    // knight offsets (d_rank, d_file) in the order of tools.py:238-247
    constexpr int kn[8][2] = {{-2, -1}, {-1, -2}, {-2, 1}, {1, -2}, {2, -1}, {-1, 2}, {2, 1}, {1, 2}};
    for (int r = 0; r < 10; ++r) {
        for (int f = 0; f < 9; ++f) {
            const int src = f + 9 * r;
            for (int rr = 0; rr < 10; ++rr) {
                if (rr == r) continue;
                t.from[n] = (uint8_t)src; t.to[n] = (uint8_t)(f + 9 * rr); ++n;
            }
            for (int ff = 0; ff < 9; ++ff) {
                if (ff == f) continue;
                t.from[n] = (uint8_t)src; t.to[n] = (uint8_t)(ff + 9 * r); ++n;
            }
            for (int j = 0; j < 8; ++j) {
                const int rr = r + kn[j][0], ff = f + kn[j][1];
                if (rr < 0 || rr > 9 || ff < 0 || ff > 8) continue;
                t.from[n] = (uint8_t)src; t.to[n] = (uint8_t)(ff + 9 * rr); ++n;
            }
        }
    }
    // advisor moves (tools.py:178-195): (file,rank) pairs, both directions of each palace diagonal
    constexpr int adv[16][4] = {
        {3, 0, 4, 1}, {4, 1, 3, 0}, {5, 0, 4, 1}, {4, 1, 5, 0}, {3, 2, 4, 1}, {4, 1, 3, 2}, {5, 2, 4, 1}, {4, 1, 5, 2},
        {3, 9, 4, 8}, {4, 8, 3, 9}, {5, 9, 4, 8}, {4, 8, 5, 9}, {3, 7, 4, 8}, {4, 8, 3, 7}, {5, 7, 4, 8}, {4, 8, 5, 7}};
    for (int j = 0; j < 16; ++j) {
        t.from[n] = (uint8_t)(adv[j][0] + 9 * adv[j][1]); t.to[n] = (uint8_t)(adv[j][2] + 9 * adv[j][3]); ++n;
    }
    // bishop moves (tools.py:197-230)
    constexpr int bis[32][4] = {
        {0, 2, 2, 0}, {2, 0, 0, 2}, {0, 2, 2, 4}, {2, 4, 0, 2}, {2, 0, 4, 2}, {4, 2, 2, 0}, {2, 4, 4, 2}, {4, 2, 2, 4},
        {4, 2, 6, 0}, {6, 0, 4, 2}, {4, 2, 6, 4}, {6, 4, 4, 2}, {6, 0, 8, 2}, {8, 2, 6, 0}, {6, 4, 8, 2}, {8, 2, 6, 4},
        {0, 7, 2, 5}, {2, 5, 0, 7}, {0, 7, 2, 9}, {2, 9, 0, 7}, {2, 5, 4, 7}, {4, 7, 2, 5}, {2, 9, 4, 7}, {4, 7, 2, 9},
        {4, 7, 6, 5}, {6, 5, 4, 7}, {4, 7, 6, 9}, {6, 9, 4, 7}, {6, 5, 8, 7}, {8, 7, 6, 5}, {6, 9, 8, 7}, {8, 7, 6, 9}};
    for (int j = 0; j < 32; ++j) {
        t.from[n] = (uint8_t)(bis[j][0] + 9 * bis[j][1]); t.to[n] = (uint8_t)(bis[j][2] + 9 * bis[j][3]); ++n;
    }
    for (int i = 0; i < kNMoves; ++i) t.inv[t.from[i] * kNSq + t.to[i]] = (uint16_t)i;
    for (int i = 0; i < kNMoves; ++i) {
        const int fr = t.from[i], to = t.to[i];
        const int mf = (8 - fr % 9) + 9 * (fr / 9), mt = (8 - to % 9) + 9 * (to / 9);
        t.flip[i] = t.inv[mf * kNSq + mt];
    }
    return t;
}

} // namespace ccz
