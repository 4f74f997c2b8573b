/*
 * bsum_hip.c -- a `bee2cmd bsum`-style front-end on the batch API.
 *
 * bee2cmd bsum hashes its files one after another, each through a bashHashStepH loop
 * (cmd/bsum/bsum.c:133-221).  This front-end reads all files, packs them back to back and hashes
 * them in ONE launch through bee2hip_hash_ragged().  Output format is bsum's:
 *     HEX(hash)  file_name
 *
 *   cc -Iinclude examples/bsum_hip.c -Lbee2_amd/lib -lbee2hip -Wl,-rpath,$PWD/bee2_amd/lib -o bsum_hip
 *   ./bsum_hip [-belt-hash | -bash256 | -bash384 | -bash512] file...
 *   (-bashNNN = bashHashStart(state, NNN / 2), as bsum.c:152-155)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bee2hip.h"

int main(int argc, char **argv)
{
    size_t alg = 0;                                 /* belt-hash by default (bsum.c:392-394) */
    int first = 1;
    if (argc > 1 && argv[1][0] == '-') {
        if (!strcmp(argv[1], "-belt-hash")) alg = 0;
        else if (!strcmp(argv[1], "-bash256")) alg = 128;
        else if (!strcmp(argv[1], "-bash384")) alg = 192;
        else if (!strcmp(argv[1], "-bash512")) alg = 256;
        else { fprintf(stderr, "usage: %s [-belt-hash|-bash256|-bash384|-bash512] file...\n", argv[0]); return 2; }
        first = 2;
    }
    const size_t n = (size_t)(argc - first);
    if (!n) return 0;
    uint64_t *off = (uint64_t *)calloc(n + 1, sizeof *off);
    octet *data = NULL;
    size_t cap = 0;
    for (size_t i = 0; i < n; ++i) {
        FILE *f = fopen(argv[first + i], "rb");
        if (!f) { printf("%s: FAILED [open]\n", argv[first + i]); free(off); free(data); return 1; }
        for (;;) {
            if (cap - off[i + 1] < (1u << 16) || !data) {
                cap = cap ? 2 * cap : (1u << 20);
                data = (octet *)realloc(data, cap);
                if (!data) return 1;
            }
            if (off[i + 1] < off[i]) off[i + 1] = off[i];
            const size_t got = fread(data + off[i + 1], 1, cap - off[i + 1], f);
            off[i + 1] += got;
            if (!got) break;
        }
        fclose(f);
        if (i + 1 < n) off[i + 2] = off[i + 1];
    }
    const size_t dlen = alg ? alg / 4 : 32;
    octet *dig = (octet *)malloc(n * dlen);
    const err_t code = bee2hip_hash_ragged(alg, data, off, n, dig);
    if (code != ERR_OK) { fprintf(stderr, "bsum_hip: err %u %s\n", code, bee2hip_last_error()); return 1; }
    for (size_t i = 0; i < n; ++i) {
        for (size_t k = 0; k < dlen; ++k) printf("%02X", dig[i * dlen + k]);
        printf("  %s\n", argv[first + i]);
    }
    free(dig); free(data); free(off);
    return 0;
}
