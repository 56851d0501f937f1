"""Prints SDRX_SINCOS_TAB_VALUES for include/sdrx_math.h: sin and cos of k*pi/32, k = 0 .. 63, evaluated in x87 extended
precision (64-bit significand) and rounded to double, as C hexadecimal floating constants (exact on every compiler).
The symmetries of the circle are imposed on the table (sin(pi - a) = sin a, cos a = sin(pi/2 - a), ...), so that e.g.
sin(pi) is exactly 0 and cos(pi/2) is exactly 0."""
import numpy as np

ld = np.longdouble
assert np.finfo(ld).nmant >= 63, "needs x87 extended precision"
PI = ld("3.14159265358979323846264338327950288419716939937510")
first = [float(np.sin(ld(k) * PI / ld(32))) for k in range(17)]  # 0 .. pi/2
first[0], first[16] = 0.0, 1.0


def s(k):
    k %= 64
    if k <= 16:
        return first[k]
    if k <= 32:
        return first[32 - k]
    return -s(k - 32)


def c(k):
    return s(k + 16)


rows = []
for k in range(64):
    rows.append("    %s, %s, /* k = %2d */ \\" % (float(s(k)).hex(), float(c(k)).hex(), k))
print("#define SDRX_SINCOS_TAB_VALUES \\")
print("\n".join(rows).rstrip(" \\").rstrip(","))
