/* TEST INFRASTRUCTURE.  A stand-in for the two ROCTx entry points libtrpl_hip.so binds at run time (csrc/trpl_api.hip
 * roctx(): dlsym(RTLD_DEFAULT, ...) first, the way `rocprofv3 --marker-trace` makes them visible by preloading
 * librocprofiler-sdk-roctx.so).  Preloaded into a child process by tests/test_abi.py; it records what was pushed. */
#include <string.h>

static int depth = 0, pushes = 0, pops = 0, max_depth = 0;
static char names[16][96];

int roctxRangePushA(const char *name)
{
    if (pushes < 16) {
        strncpy(names[pushes], name ? name : "", 95);
        names[pushes][95] = 0;
    }
    pushes++;
    depth++;
    if (depth > max_depth) max_depth = depth;
    return depth - 1;
}

int roctxRangePop(void)
{
    pops++;
    return --depth;
}

int mock_roctx_pushes(void) { return pushes; }
int mock_roctx_pops(void) { return pops; }
int mock_roctx_depth(void) { return depth; }
int mock_roctx_max_depth(void) { return max_depth; }
const char *mock_roctx_name(int i) { return i >= 0 && i < 16 && i < pushes ? names[i] : ""; }
