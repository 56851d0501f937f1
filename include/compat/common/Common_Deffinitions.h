#include <sdrx/common.h>
