#include <sdrx/AM.h>
