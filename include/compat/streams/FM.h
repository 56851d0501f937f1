#include <sdrx/FM.h>
