/*
 * bsum_hip.c -- a `bee2cmd bsum`-style front-end on the batch API.
 *
 * bee2cmd bsum hashes its files one after another, each through a bashHashStepH / beltHashStepH loop
 * (cmd/bsum/bsum.c:133-221).  This front-end reads all files, packs them back to back and hashes them in ONE
 * launch per device through bee2hip_hash_ragged_multi() (every visible GPU takes a byte-balanced range of files).
 *
 *   print mode (bsum.c:207-224):   HEX(hash)  file_name        -- lower-case hex, two spaces, as bsumPrint
 *   check mode (bsum.c:226-306):   -c sums_file                -- every line "hex  name" of sums_file is checked:
 *        "name: OK" / "name: FAILED [checksum]" / "name: FAILED [open]" on stdout, bsum's WARNING lines on stderr,
 *        exit status -1 (255) if any line, file or checksum was bad.  All listed files are hashed in one batch.
 *
 *   cc -Iinclude examples/bsum_hip.c -Lbee2_amd/lib -lbee2hip -Wl,-rpath,$PWD/bee2_amd/lib -o bsum_hip
 *   ./bsum_hip [-belt-hash | -bash256 | -bash384 | -bash512] file...
 *   ./bsum_hip [-belt-hash | -bash256 | -bash384 | -bash512] -c sums_file
 *   (-bashNNN = bashHashStart(state, NNN / 2), as bsum.c:152-155)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bee2hip.h"

typedef struct { uint64_t *off; octet *data; size_t cap, n; } pack_t;

/* append a file to the pack; returns 0, or -1 when it cannot be opened / read (nothing appended) */
static int pack_file(pack_t *p, const char *name)
{
    FILE *f = fopen(name, "rb");
    if (!f) return -1;
    const uint64_t start = p->off[p->n];
    uint64_t end = start;
    for (;;) {
        if (!p->data || p->cap - end < (1u << 16)) {
            p->cap = p->cap ? 2 * p->cap : (1u << 20);
            p->data = (octet *)realloc(p->data, p->cap);
            if (!p->data) { fclose(f); exit(1); }
        }
        const size_t got = fread(p->data + end, 1, p->cap - end, f);
        end += got;
        if (!got) break;
    }
    const int bad = ferror(f);
    fclose(f);
    if (bad) return -1;
    p->off[++p->n] = end;
    return 0;
}

static int hexval(int c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

static int check_mode(size_t alg, const char *sums)
{
    const size_t dlen = alg ? alg / 4 : 32;
    FILE *f = fopen(sums, "rb");
    if (!f) { printf("%s: No such file\n", sums); return -1; }
    char line[1024];
    size_t all_lines = 0, bad_lines = 0, bad_files = 0, bad_hashes = 0, cap = 64, m = 0;
    char **names = (char **)malloc(cap * sizeof *names);
    octet *want = (octet *)malloc(cap * dlen);
    pack_t p = {(uint64_t *)calloc(cap + 1, sizeof(uint64_t)), NULL, 0, 0};
    for (; fgets(line, sizeof line, f); ++all_lines) {
        size_t len = strlen(line);
        int ok = len >= 2 * dlen + 2 && line[2 * dlen] == ' ' && line[2 * dlen + 1] == ' ';
        for (size_t k = 0; ok && k < 2 * dlen; ++k) ok = hexval((unsigned char)line[k]) >= 0;
        if (!ok) { bad_lines++; continue; }
        if (line[len - 1] == '\n') line[--len] = 0;
        if (len && line[len - 1] == '\r') line[--len] = 0;
        if (m == cap) {
            cap *= 2;
            names = (char **)realloc(names, cap * sizeof *names);
            want = (octet *)realloc(want, cap * dlen);
            p.off = (uint64_t *)realloc(p.off, (cap + 1) * sizeof(uint64_t));
        }
        const char *name = line + 2 * dlen + 2;
        if (pack_file(&p, name) != 0) { printf("%s: FAILED [open]\n", name); bad_files++; continue; }
        for (size_t k = 0; k < dlen; ++k)
            want[m * dlen + k] = (octet)(hexval((unsigned char)line[2 * k]) << 4 | hexval((unsigned char)line[2 * k + 1]));
        names[m] = strdup(name);
        ++m;
    }
    fclose(f);
    octet *dig = (octet *)malloc(m ? m * dlen : 1);
    const err_t code = bee2hip_hash_ragged_multi(alg, p.data, p.off, m, dig, 0);
    if (code != ERR_OK) { fprintf(stderr, "bsum_hip: err %u %s\n", code, bee2hip_last_error()); return 1; }
    for (size_t i = 0; i < m; ++i) {
        if (memcmp(dig + i * dlen, want + i * dlen, dlen)) { bad_hashes++; printf("%s: FAILED [checksum]\n", names[i]); }
        else printf("%s: OK\n", names[i]);
        free(names[i]);
    }
    if (bad_lines)
        fprintf(stderr, bad_lines == 1 ? "WARNING: %lu input line (out of %lu) is improperly formatted\n"
                                       : "WARNING: %lu input lines (out of %lu) are improperly formatted\n",
                (unsigned long)bad_lines, (unsigned long)all_lines);
    if (bad_files)
        fprintf(stderr, bad_files == 1 ? "WARNING: %lu listed file could not be opened or read\n"
                                       : "WARNING: %lu listed files could not be opened or read\n", (unsigned long)bad_files);
    if (bad_hashes)
        fprintf(stderr, bad_hashes == 1 ? "WARNING: %lu computed checksum did not match\n"
                                        : "WARNING: %lu computed checksums did not match\n", (unsigned long)bad_hashes);
    free(dig); free(want); free(names); free(p.data); free(p.off);
    return (bad_lines || bad_files || bad_hashes) ? -1 : 0;
}

int main(int argc, char **argv)
{
    size_t alg = 0;                                 /* belt-hash by default (bsum.c:392-394) */
    int first = 1;
    if (argc > 1 && argv[1][0] == '-' && strcmp(argv[1], "-c")) {
        if (!strcmp(argv[1], "-belt-hash")) alg = 0;
        else if (!strcmp(argv[1], "-bash256")) alg = 128;
        else if (!strcmp(argv[1], "-bash384")) alg = 192;
        else if (!strcmp(argv[1], "-bash512")) alg = 256;
        else { fprintf(stderr, "usage: %s [-belt-hash|-bash256|-bash384|-bash512] [-c sums_file | file...]\n", argv[0]); return 2; }
        first = 2;
    }
    if (argc > first && !strcmp(argv[first], "-c")) {
        if (argc != first + 2) { fprintf(stderr, "usage: %s [alg] -c sums_file\n", argv[0]); return 2; }
        return check_mode(alg, argv[first + 1]);
    }
    const size_t n = (size_t)(argc - first);
    if (!n) return 0;
    pack_t p = {(uint64_t *)calloc(n + 1, sizeof(uint64_t)), NULL, 0, 0};
    int ret = 0;
    int *slot = (int *)malloc(n * sizeof *slot);    /* file i -> index in the pack, or -1 */
    for (size_t i = 0; i < n; ++i) {
        slot[i] = (int)p.n;
        if (pack_file(&p, argv[first + i]) != 0) { printf("%s: FAILED [open]\n", argv[first + i]); slot[i] = -1; ret = -1; }
    }
    const size_t dlen = alg ? alg / 4 : 32;
    octet *dig = (octet *)malloc(p.n ? p.n * dlen : 1);
    const err_t code = bee2hip_hash_ragged_multi(alg, p.data, p.off, p.n, dig, 0);
    if (code != ERR_OK) { fprintf(stderr, "bsum_hip: err %u %s\n", code, bee2hip_last_error()); return 1; }
    for (size_t i = 0; i < n; ++i) {
        if (slot[i] < 0) continue;
        for (size_t k = 0; k < dlen; ++k) printf("%02x", dig[(size_t)slot[i] * dlen + k]);
        printf("  %s\n", argv[first + i]);
    }
    free(dig); free(p.data); free(p.off); free(slot);
    return ret;
}
