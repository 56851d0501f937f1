#include <sdrx/QPSK.h>
