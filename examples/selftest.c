/* selftest.c -- the STB known-answer vectors through libbee2hip's bee2 drop-in functions, from plain C.
 *
 *   cc -Iinclude examples/selftest.c -Lbee2_amd/lib -lbee2hip -Wl,-rpath,bee2_amd/lib -o selftest
 *   ./selftest vectors.txt
 *
 * vectors.txt is line based (tests/test_gpu_mixed.py writes it from tests/golden/stb_kat.json, belt_dwp.json,
 * belt_che.json -- the vectors bee2's own tests hold, test/crypto/{bash,belt,bign}_test.c):
 *   bashF    <name> <in> <out>
 *   bashhash <name> <l> <msg> <out>
 *   belthash <name> <msg> <out>
 *   ctr      <name> <in> <key> <iv> <out>
 *   mac      <name> <in> <key> <out>
 *   mode     <name> <fn> <in> <key> <iv|-> <out>          fn = beltECBEncr .. beltSDEDecr
 *   wrap     <name> <DWP|CHE> <crit> <open> <key> <iv> <out> <mac>
 *   verify   <name> <hash> <sig> <pubkey> <code>
 * "-" stands for an empty octet string.  Prints one line per vector, exit status = number of failures. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bee2hip.h"

#define MAXF 9
#define MAXB 4096

static size_t unhex(octet *dst, const char *s)
{
    size_t n = 0;
    if (strcmp(s, "-") == 0) return 0;
    for (; s[0] && s[1]; s += 2) {
        unsigned v;
        sscanf(s, "%2x", &v);
        dst[n++] = (octet)v;
    }
    return n;
}
static int same(const octet *a, const octet *b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; }

typedef err_t (*mode4_fn)(void *, const void *, size_t, const octet[], size_t);
typedef err_t (*mode5_fn)(void *, const void *, size_t, const octet[], size_t, const octet[16]);

int main(int argc, char **argv)
{
    static char line[1 << 16];
    static octet f[MAXF][MAXB], out[MAXB];
    size_t n[MAXF];
    int fails = 0, total = 0;
    FILE *fp = argc > 1 ? fopen(argv[1], "r") : NULL;
    if (!fp) { fprintf(stderr, "usage: selftest vectors.txt\n"); return 255; }
    while (fgets(line, sizeof line, fp)) {
        char *tok[MAXF + 3];
        int nt = 0, ok = 0;
        for (char *p = strtok(line, " \t\r\n"); p && nt < MAXF + 3; p = strtok(NULL, " \t\r\n")) tok[nt++] = p;
        if (nt < 3 || tok[0][0] == '#') continue;
        const char *op = tok[0], *name = tok[1];
        if (!strcmp(op, "bashF") && nt == 4) {
            unhex(f[0], tok[2]); unhex(f[1], tok[3]);
            bashF(f[0], NULL);
            ok = same(f[0], f[1], 192);
        } else if (!strcmp(op, "bashhash") && nt == 5) {
            size_t l = (size_t)atoi(tok[2]);
            n[0] = unhex(f[0], tok[3]); n[1] = unhex(f[1], tok[4]);
            ok = bashHash(out, l, f[0], n[0]) == ERR_OK && same(out, f[1], n[1]);
        } else if (!strcmp(op, "belthash") && nt == 4) {
            n[0] = unhex(f[0], tok[2]); unhex(f[1], tok[3]);
            ok = beltHash(out, f[0], n[0]) == ERR_OK && same(out, f[1], 32);
        } else if (!strcmp(op, "ctr") && nt == 6) {
            n[0] = unhex(f[0], tok[2]); n[1] = unhex(f[1], tok[3]); unhex(f[2], tok[4]); unhex(f[3], tok[5]);
            ok = beltCTR(out, f[0], n[0], f[1], n[1], f[2]) == ERR_OK && same(out, f[3], n[0]);
        } else if (!strcmp(op, "mac") && nt == 5) {
            n[0] = unhex(f[0], tok[2]); n[1] = unhex(f[1], tok[3]); unhex(f[2], tok[4]);
            ok = beltMAC(out, f[0], n[0], f[1], n[1]) == ERR_OK && same(out, f[2], 8);
        } else if (!strcmp(op, "mode") && nt == 7) {
            const char *fn = tok[2];
            n[0] = unhex(f[0], tok[3]); n[1] = unhex(f[1], tok[4]); n[2] = unhex(f[2], tok[5]); unhex(f[3], tok[6]);
            err_t rc = 1;
            static const struct { const char *nm; mode4_fn f4; mode5_fn f5; } T[] = {
                {"beltECBEncr", beltECBEncr, NULL}, {"beltECBDecr", beltECBDecr, NULL},
                {"beltCBCEncr", NULL, beltCBCEncr}, {"beltCBCDecr", NULL, beltCBCDecr},
                {"beltBDEEncr", NULL, beltBDEEncr}, {"beltBDEDecr", NULL, beltBDEDecr},
                {"beltSDEEncr", NULL, beltSDEEncr}, {"beltSDEDecr", NULL, beltSDEDecr}};
            for (size_t i = 0; i < sizeof T / sizeof T[0]; ++i)
                if (!strcmp(fn, T[i].nm))
                    rc = T[i].f4 ? T[i].f4(out, f[0], n[0], f[1], n[1]) : T[i].f5(out, f[0], n[0], f[1], n[1], f[2]);
            ok = rc == ERR_OK && same(out, f[3], n[0]);
        } else if (!strcmp(op, "wrap") && nt == 9) {
            octet mac[8];
            n[0] = unhex(f[0], tok[3]); n[1] = unhex(f[1], tok[4]); n[2] = unhex(f[2], tok[5]); unhex(f[3], tok[6]);
            unhex(f[4], tok[7]); unhex(f[5], tok[8]);
            err_t rc = !strcmp(tok[2], "DWP") ? beltDWPWrap(out, mac, f[0], n[0], f[1], n[1], f[2], n[2], f[3])
                                               : beltCHEWrap(out, mac, f[0], n[0], f[1], n[1], f[2], n[2], f[3]);
            ok = rc == ERR_OK && same(out, f[4], n[0]) && same(mac, f[5], 8);
            if (ok) {                                   /* and back: the tag verifies, a wrong one is refused */
                octet back[MAXB];
                err_t (*un)(void *, const void *, size_t, const void *, size_t, const octet[8], const octet[], size_t,
                            const octet[16]) = !strcmp(tok[2], "DWP") ? beltDWPUnwrap : beltCHEUnwrap;
                ok = un(back, out, n[0], f[1], n[1], mac, f[2], n[2], f[3]) == ERR_OK && same(back, f[0], n[0]);
                mac[0] ^= 1;
                ok = ok && un(back, out, n[0], f[1], n[1], mac, f[2], n[2], f[3]) == ERR_BAD_MAC;
            }
        } else if (!strcmp(op, "verify") && nt == 6) {
            unhex(f[0], tok[2]); unhex(f[1], tok[3]); unhex(f[2], tok[4]);
            ok = bign128Verify(f[0], f[1], f[2]) == (err_t)atoi(tok[5]);
        } else {
            printf("SKIP %s %s\n", op, name);
            continue;
        }
        ++total;
        if (!ok) ++fails;
        printf("%s %s %s\n", ok ? "OK  " : "FAIL", op, name);
    }
    fclose(fp);
    printf("%d vectors, %d failed (%s)\n", total, fails, bash_platform);
    return fails;
}
