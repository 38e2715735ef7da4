/*
 * check_tables_ref.c -- test infrastructure (see mc_oracle.h).
 *
 * Compiles the reference's own table header IN PLACE
 * (/root/reference/Source/marching_lookup.h, a self-contained header: no
 * windows.h, no stand-ins needed) and checks that this repo's packed tables
 * (include/mc_tables_data.h) decode to exactly the same entries.  Built by
 * oracle/Makefile into oracle/_ref/ only where /root/reference exists.
 */
#include <stdint.h>
#include <stdio.h>

#include "marching_lookup.h" /* found through -I/root/reference/Source */
#include "mc_oracle.h"

int main(void) {
    int bad = 0;
    const uint64_t *rows = orc_tri_rows();
    for (int c = 0; c < 256; c++) {
        int n = 0;
        for (int k = 0; k < 16; k++) {
            int nib = (int)((rows[c] >> (4 * k)) & 0xF);
            int e = nib == 0xF ? -1 : nib;
            if (e != tri_table[c][k]) bad++;
            if (tri_table[c][k] >= 0) n++;
        }
        if (n / 3 != orc_tri_counts()[c]) bad++;
        int f = orc_amb_faces()[c];
        if (f == 0xFF) {
            for (int k = 0; k < 5; k++)
                if (ambiguity_check_and_redirect[c][k] != -1) bad++;
        } else {
            if (ambiguity_check_and_redirect[c][0] != 255 - c) bad++;
            for (int k = 0; k < 4; k++)
                if (ambiguity_check_and_redirect[c][k + 1] != ((orc_face_corners()[f] >> (4 * k)) & 0xF)) bad++;
        }
    }
    for (int f = 0; f < 6; f++)
        for (int k = 0; k < 4; k++)
            if (cube_face_vertex_table[f][k] != ((orc_face_corners()[f] >> (4 * k)) & 0xF)) bad++;
    for (int e = 0; e < 12; e++) {
        if (cube_edge_vertex_table[e][0] != (orc_edge_corners()[e] & 0xF)) bad++;
        if (cube_edge_vertex_table[e][1] != (orc_edge_corners()[e] >> 4)) bad++;
    }
    for (int i = 0; i < 8; i++)
        if (two_to_the[i] != (1 << i)) bad++;
    printf("check_tables_ref: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
