/*
 * sigvfy_hip.c -- the batch shape of `bee2cmd sig vfy` (cmd/core/cmd_sig.c:461-490): for every
 * file, hash it, validate the signer's public key, verify the signature -- here for MANY files with three
 * launches on one stream (ragged belt-hash -> bign128PubkeyVal -> bign128Verify), digests never leaving the GPU;
 * a list under ONE public key takes the one-signer entry (bee2hip_bignVerifyL_onekey_batch_dev), one of a few signers the keyed one.
 *
 * Input: a list file, one line per signed file:   <file name>  <hex signature, 48 octets>  <hex public key, 64 octets>
 * (bee2cmd keeps signature and certificate chain in a DER container appended to the file, cmd_sig.c:121-330;
 *  parsing that container is PKI plumbing outside this library's scope -- the three primitive calls per file
 *  and their order are what this example reproduces.)
 * Output, per line: "<name>: OK" | "<name>: FAILED [open]" | "<name>: FAILED [pubkey]" | "<name>: FAILED [signature]";
 * exit status 0 iff every line verified.
 *
 *   cc -Iinclude examples/sigvfy_hip.c -Lbee2_amd/lib -lbee2hip -L/opt/rocm/lib -lamdhip64 \
 *      -Wl,-rpath,$PWD/bee2_amd/lib -Wl,-rpath,/opt/rocm/lib -o sigvfy_hip
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bee2hip.h"

/* the three HIP runtime calls this example needs (device buffers); declared here so that no HIP header is required */
extern int hipMalloc(void **ptr, size_t size);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind);   /* 1 = H2D, 2 = D2H */
extern int hipFree(void *ptr);

static int hexval(int c)
{
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
static int unhex(octet *out, size_t n, const char *s)
{
    if (strlen(s) != 2 * n) return -1;
    for (size_t i = 0; i < n; ++i) {
        const int a = hexval((unsigned char)s[2 * i]), b = hexval((unsigned char)s[2 * i + 1]);
        if (a < 0 || b < 0) return -1;
        out[i] = (octet)(a << 4 | b);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s list_file\n", argv[0]); return 2; }
    FILE *lf = fopen(argv[1], "r");
    if (!lf) { fprintf(stderr, "%s: cannot open\n", argv[1]); return 2; }
    size_t cap = 64, n = 0, dcap = 1 << 20, failed = 0;
    char **names = (char **)malloc(cap * sizeof *names);
    octet *sigs = (octet *)malloc(cap * 48), *pubs = (octet *)malloc(cap * 64), *data = (octet *)malloc(dcap);
    uint64_t *off = (uint64_t *)calloc(cap + 1, sizeof *off);
    char name[600], sh[200], ph[300];
    while (fscanf(lf, "%599s %199s %299s", name, sh, ph) == 3) {
        if (n == cap) {
            cap *= 2;
            names = (char **)realloc(names, cap * sizeof *names);
            sigs = (octet *)realloc(sigs, cap * 48); pubs = (octet *)realloc(pubs, cap * 64);
            off = (uint64_t *)realloc(off, (cap + 1) * sizeof *off);
        }
        FILE *f = fopen(name, "rb");
        if (!f || unhex(sigs + 48 * n, 48, sh) || unhex(pubs + 64 * n, 64, ph)) {
            printf("%s: FAILED [open]\n", name); failed++;
            if (f) fclose(f);
            continue;
        }
        uint64_t end = off[n];
        for (;;) {
            if (dcap - end < (1u << 16)) { dcap *= 2; data = (octet *)realloc(data, dcap); }
            const size_t got = fread(data + end, 1, dcap - end, f);
            end += got;
            if (!got) break;
        }
        fclose(f);
        names[n] = strdup(name);
        off[++n] = end;
    }
    fclose(lf);
    if (n) {
        if (bee2hip_set_device(0) != ERR_OK) { fprintf(stderr, "sigvfy_hip: %s\n", bee2hip_last_error()); return 1; }
        void *d_data, *d_off, *d_hash, *d_sig, *d_pub, *d_c1, *d_c2;
        const size_t total = (size_t)off[n];
        if (hipMalloc(&d_data, total + 16) || hipMalloc(&d_off, (n + 1) * 8) || hipMalloc(&d_hash, n * 32) || hipMalloc(&d_sig, n * 48) ||
            hipMalloc(&d_pub, n * 64) || hipMalloc(&d_c1, n * 4) || hipMalloc(&d_c2, n * 4)) { fprintf(stderr, "sigvfy_hip: out of device memory\n"); return 1; }
        hipMemcpy(d_data, data, total, 1); hipMemcpy(d_off, off, (n + 1) * 8, 1);
        hipMemcpy(d_sig, sigs, n * 48, 1); hipMemcpy(d_pub, pubs, n * 64, 1);
        /* hash (belt-hash, bign128's pre-hash) -> key validation -> verification: one stream, no host round trip */
        err_t code = bee2hip_hash_ragged_dev(0, d_data, d_off, n, d_hash, NULL);
        if (code == ERR_OK) code = bee2hip_bignPubkeyValL_batch_dev(128, d_pub, n, d_c1, NULL);
        /* the signers of the list: ONE key (the usual tree of files under one key) or a FEW take the entries that treat a key as a
           fixed base with a comb table of its own -- a fifth to a quarter of the general entry's work, same verdicts; a crowd of
           signers (more than 256 here) takes the general entry */
        enum { MAX_SIGNERS = 256 };
        octet *keys = (octet *)malloc((size_t)MAX_SIGNERS * 64);
        uint32_t *kidx = (uint32_t *)malloc(n * sizeof *kidx);
        size_t nkeys = 0;
        for (size_t i = 0; i < n && nkeys <= MAX_SIGNERS; ++i) {
            size_t k = 0;
            while (k < nkeys && memcmp(keys + 64 * k, pubs + 64 * i, 64)) ++k;
            if (k == nkeys) { if (nkeys == MAX_SIGNERS) { nkeys++; break; } memcpy(keys + 64 * nkeys++, pubs + 64 * i, 64); }
            kidx[i] = (uint32_t)k;
        }
        static const octet oid_belt_hash[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};
        if (code == ERR_OK) {
            if (nkeys == 1)
                code = bee2hip_bignVerifyL_onekey_batch_dev(128, oid_belt_hash, sizeof oid_belt_hash, d_hash, d_sig, keys, n, d_c2, NULL);
            else if (nkeys <= MAX_SIGNERS) {
                hipMemcpy(d_pub, kidx, n * 4, 1);         /* the key indices take the place of the keys on the device */
                code = bee2hip_bignVerifyL_keyed_batch_dev(128, oid_belt_hash, sizeof oid_belt_hash, d_hash, d_sig, keys, nkeys, d_pub, n, d_c2, NULL);
            } else
                code = bee2hip_bign128Verify_batch_dev(d_hash, d_sig, d_pub, n, d_c2, NULL);
        }
        free(keys); free(kidx);
        if (code != ERR_OK) { fprintf(stderr, "sigvfy_hip: err %u %s\n", code, bee2hip_last_error()); return 1; }
        err_t *c1 = (err_t *)malloc(n * 4), *c2 = (err_t *)malloc(n * 4);
        hipMemcpy(c1, d_c1, n * 4, 2); hipMemcpy(c2, d_c2, n * 4, 2);
        for (size_t i = 0; i < n; ++i) {
            if (c1[i] != ERR_OK) { printf("%s: FAILED [pubkey]\n", names[i]); failed++; }
            else if (c2[i] != ERR_OK) { printf("%s: FAILED [signature]\n", names[i]); failed++; }
            else printf("%s: OK\n", names[i]);
            free(names[i]);
        }
        free(c1); free(c2);
        hipFree(d_data); hipFree(d_off); hipFree(d_hash); hipFree(d_sig); hipFree(d_pub); hipFree(d_c1); hipFree(d_c2);
    }
    free(names); free(sigs); free(pubs); free(data); free(off);
    return failed ? 1 : 0;
}
