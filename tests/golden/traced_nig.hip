// traced by mhx.trace (advancedmh.jl_amd/mhx/trace.py): 32 operations in the source
MHX_LOGDENSITY(x, d, data, ndata)
{
    const mhx_real t1 = x[0];
    const mhx_real t2 = x[1];
    const mhx_real t3 = mhx_sqrt(t1);
    const mhx_real t4 = mhx_log(t1);
    const mhx_real t6 = MHX_R(0x1.8000000000000p+1) * t4;
    const mhx_real t8 = MHX_R(0x1.193ea7aad030bp+1) - t6;
    const mhx_real t9 = MHX_R(0x1.8000000000000p+1) / t1;
    const mhx_real t10 = t8 - t9;
    const mhx_real t11 = t2 - MHX_R(0x0.0p+0);
    const mhx_real t12 = t11 / t3;
    const mhx_real t13 = t12 * t12;
    const mhx_real t15 = t13 + MHX_R(0x1.d67f1c864beb4p+0);
    const mhx_real t16 = -t15;
    const mhx_real t18 = t16 / MHX_R(0x1.0000000000000p+1);
    const mhx_real t19 = mhx_log(t3);
    const mhx_real t20 = t18 - t19;
    const mhx_real t22 = MHX_R(0x1.8000000000000p+0) - t2;
    const mhx_real t23 = t22 / t3;
    const mhx_real t24 = t23 * t23;
    const mhx_real t25 = t24 + MHX_R(0x1.d67f1c864beb4p+0);
    const mhx_real t26 = -t25;
    const mhx_real t27 = t26 / MHX_R(0x1.0000000000000p+1);
    const mhx_real t28 = t27 - t19;
    const mhx_real t29 = MHX_R(0x1.0000000000000p+1) - t2;
    const mhx_real t30 = t29 / t3;
    const mhx_real t31 = t30 * t30;
    const mhx_real t32 = t31 + MHX_R(0x1.d67f1c864beb4p+0);
    const mhx_real t33 = -t32;
    const mhx_real t34 = t33 / MHX_R(0x1.0000000000000p+1);
    const mhx_real t35 = t34 - t19;
    const mhx_real t36 = t10 + t20;
    const mhx_real t37 = t36 + t28;
    const mhx_real t38 = t37 + t35;
    const mhx_real t40 = (t1 > MHX_R(0x0.0p+0)) ? t38 : -MHX_INF;
    return t40;
}
