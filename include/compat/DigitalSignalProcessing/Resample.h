#include <sdrx/Resample.h>
