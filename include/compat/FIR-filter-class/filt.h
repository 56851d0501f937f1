#include <sdrx/filt.h>
