// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): 12 operations in the source
MHX_LOGDENSITY(x, d, data, ndata)
{
    const mhx_real t1 = x[0];
    const mhx_real t2 = x[1];
    const mhx_real t4 = t1 / MHX_R(0x1.94c583ada5b53p+1);
    const mhx_real t6 = MHX_R(0x1.94c583ada5b52p+0) * t4;
    const mhx_real t7 = t2 - t6;
    const mhx_real t9 = t7 / MHX_R(0x1.5e8add236a58fp+1);
    const mhx_real t10 = t4 * t4;
    const mhx_real t11 = t9 * t9;
    const mhx_real t12 = t10 + t11;
    const mhx_real t13 = -t12;
    const mhx_real t15 = t13 / MHX_R(0x1.0000000000000p+1);
    const mhx_real t17 = t15 - MHX_R(0x1.26bb1bbb55516p+0);
    const mhx_real t19 = t17 - MHX_R(0x1.01e85798eb9a3p+0);
    const mhx_real t21 = t19 - MHX_R(0x1.d67f1c864beb4p+0);
    return t21;
}
