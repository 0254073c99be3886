/* gen_costable.c — regenerates the cosine table of the oracle from the reference's expression and compares.
 *
 *   COS[x][u] = cosf(((2*x+1) * u * PI) / 16),  PI = 3.14f      (loadjpg.cpp:108, :120: int * int * float / int, in fp32)
 *
 * evaluated with the libm of the machine it runs on.  The table in oracle/hjd_oracle.c is the contract (entry [6][7] is
 * specific to glibc 2.35's cosf; a correctly rounded cosf gives 0x1.1895b0p-1 there): this tool says whether the local libm
 * still reproduces it.  Build and run:  gcc -O0 -ffp-contract=off -o /tmp/gen_costable tools/gen_costable.c oracle/hjd_oracle.c -lm && /tmp/gen_costable
 * Prints the 64 entries as C99 hex floats; exit code 0 = identical to the oracle's table, 1 = differs (entries listed). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "../oracle/hjd_oracle.h"

int main(void)
{
    const float PI = 3.14f;
    const float *tab = hjo_cos_table();
    int bad = 0;
    for (int x = 0; x < 8; x++) {
        for (int u = 0; u < 8; u++) {
            volatile float arg = ((2 * x + 1) * u * PI) / 16; /* the reference's own expression */
            const float c = cosf(arg);
            printf("%s%a", u ? ", " : "", (double)c);
            if (memcmp(&c, &tab[x * 8 + u], sizeof c)) {
                bad++;
                fprintf(stderr, "COS[%d][%d]: this libm %a, table %a\n", x, u, (double)c, (double)tab[x * 8 + u]);
            }
        }
        printf("\n");
    }
    return bad ? 1 : 0;
}
