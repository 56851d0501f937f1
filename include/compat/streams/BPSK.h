#include <sdrx/BPSK.h>
