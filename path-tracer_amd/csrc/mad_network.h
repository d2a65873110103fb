// mad_network.h -- the reference's "median getter" procedure (DenoiserNN.cpp:190-256)
// as a fixed network of comparisons and selects.
//
// The procedure pushes nine values through a five-slot array m[0..4] (all zero at the
// start) twice.  For each value v it scans k = 1..4 and stops at the first k with
// !(v > m[k]) (the comparison with m[0] has no effect in either pass):
//   pass 1: stopped at k -> m[k-1] = v;            never stopped -> m[0..3] = m[4], m[4] = v
//   pass 2: stopped at k -> m[0..k-1] = m[k], m[k] = v;   never stopped -> as k = 4
// Between the passes every value becomes |v - m[0]|; the result is m[0] after pass 2.
// It is not a median and depends on the insertion order, so it is followed literally:
// every comparison below is one the scan makes, on the same operands, and every slot
// receives the value the scan would store.  What is left out is work whose outcome is
// known: comparisons against slots that still hold their initial zero coincide, pass 2
// never reads m[0] before its last step, and the last step of each pass only has to
// produce m[0].
//
// The code is written against a policy `O` so that the same network runs on the device
// (wave masks in scalar registers, v_cndmask selects) and on the host (bools), where it is
// checked exhaustively against the literal loop:
//   O::Mask, O::gt(a, b) -> Mask (a > b, false for NaN), O::land / O::lor / O::andn(a, b) = a & !b,
//   O::lnot, O::sel(mask, x, y) = mask ? x : y, O::fabs.
// The includer defines PTK_MAD_FN (function qualifiers) and PTK_MAD_UNROLL (loop pragma or nothing).
#pragma once

namespace ptk_mad {

template <class O>
struct Network {
    using M = typename O::Mask;

    // One general pass-1 step on slots (m0..m4).
    //
    // In pass 1 the slots m1 <= m2 <= m3 <= m4 stay in ascending order: they start equal, a
    // step that stops at k >= 2 stores a v with m[k-1] < v <= m[k] into m[k-1], a step that
    // stops at 1 writes m0 only, and a step that never stops sets m1..m3 to the old m4 and m4
    // to a larger v.  (A NaN never enters them: v > m[k] is false for it, so it stops at 1.)
    // Hence v > m[k] implies v > m[j] for every j < k, "the scan reached k" is just
    // v > m[k-1], and the conjunctions of the literal scan drop out.
    static PTK_MAD_FN void pass1_step(float v, float& m0, float& m1, float& m2, float& m3, float& m4)
    {
        const M g1 = O::gt(v, m1), g2 = O::gt(v, m2), g3 = O::gt(v, m3), g4 = O::gt(v, m4);
        const float t = O::sel(g4, m4, v);                         // what a written slot receives
        const float n4 = O::sel(g4, v, m4);
        m0 = O::sel(O::lor(g4, O::lnot(g1)), t, m0);               // stopped at 1, or shifted
        m1 = O::sel(O::lor(g4, O::andn(g1, g2)), t, m1);           // stopped at 2, or shifted
        m2 = O::sel(O::lor(g4, O::andn(g2, g3)), t, m2);
        m3 = O::sel(g3, t, m3);                                    // stopped at 4, or shifted
        m4 = n4;
    }

    // m[0] after pass 1 over v[0..8].
    static PTK_MAD_FN float pass1(const float (&v)[9])
    {
        // step 1, all slots zero: the four comparisons coincide
        const M p = O::gt(v[0], 0.0f);
        float m0 = O::sel(p, 0.0f, v[0]);
        float m4 = O::sel(p, v[0], 0.0f);
        float m1, m2, m3;
        {   // step 2, m1 = m2 = m3 = 0
            const float x = v[1];
            const M q = O::gt(x, 0.0f), g4 = O::gt(x, m4);
            const M a4 = O::land(q, g4);
            const float t = O::sel(a4, m4, x);
            const float n4 = O::sel(a4, x, m4);
            m0 = O::sel(O::lor(a4, O::lnot(q)), t, m0);
            m1 = O::sel(a4, m4, 0.0f);
            m2 = m1;
            m3 = O::sel(q, t, 0.0f);                               // stopped at 4, or shifted
            m4 = n4;
        }
        PTK_MAD_UNROLL
        for (int i = 2; i < 8; i++) pass1_step(v[i], m0, m1, m2, m3, m4);
        {   // last step: only m[0] is read afterwards
            const float x = v[8];
            const M g1 = O::gt(x, m1), g4 = O::gt(x, m4);      // ascending slots: see pass1_step
            const float t = O::sel(g4, m4, x);
            m0 = O::sel(O::lor(g4, O::lnot(g1)), t, m0);
        }
        return m0;
    }

    // One general pass-2 step on slots (m1..m4); m[0] is not tracked.
    static PTK_MAD_FN void pass2_step(float v, float& m1, float& m2, float& m3, float& m4)
    {
        const M g1 = O::gt(v, m1), g2 = O::gt(v, m2), g3 = O::gt(v, m3);
        const M a2 = O::land(g1, g2);                // stops at 3 or 4
        const M c2 = O::andn(g1, g2);                // stops at 2
        const M c3 = O::andn(a2, g3), c4 = O::land(a2, g3);
        const float hi = O::sel(g3, m4, m3);         // the stopping slot's value when a2
        const float w = O::sel(a2, hi, m2);          // ... and when it stops at 2
        const float q = O::sel(a2, hi, m3);
        m1 = O::sel(g1, w, v);                       // stopped at 1: v; else the stopping slot
        m2 = O::sel(c2, v, w);
        m3 = O::sel(c3, v, q);
        m4 = O::sel(c4, v, m4);
    }

    // m[0] after pass 2 over d[0..8] (any values; the caller passes absolute differences).
    static PTK_MAD_FN float pass2(const float (&d)[9])
    {
        // step 1, all slots zero: stops at 1 (m1 = v) unless v > 0 (m4 = v)
        const M p = O::gt(d[0], 0.0f);
        float m1 = O::sel(p, 0.0f, d[0]);
        float m4 = O::sel(p, d[0], 0.0f);
        float m2, m3;
        {   // step 2, m2 = m3 = 0: it cannot stop at 3
            const float x = d[1];
            const M g1 = O::gt(x, m1), q = O::gt(x, 0.0f);
            const M c4 = O::land(g1, q), c2 = O::andn(g1, q);
            const float y = O::sel(c4, m4, 0.0f);
            m1 = O::sel(g1, y, x);
            m2 = O::sel(c2, x, y);
            m3 = y;
            m4 = O::sel(c4, x, m4);
        }
        PTK_MAD_UNROLL
        for (int i = 2; i < 8; i++) pass2_step(d[i], m1, m2, m3, m4);
        // last step: m[0] = the stopping slot's value
        const float x = d[8];
        const M g1 = O::gt(x, m1), g2 = O::gt(x, m2), g3 = O::gt(x, m3);
        const float hi = O::sel(g3, m4, m3);
        const float w = O::sel(O::land(g1, g2), hi, m2);
        return O::sel(g1, w, m1);
    }

    static PTK_MAD_FN float run(const float (&v)[9])
    {
        const float m0 = pass1(v);
        float d[9];
        PTK_MAD_UNROLL
        for (int i = 0; i < 9; i++) d[i] = O::fabs(v[i] - m0);
        return pass2(d);
    }
};

}  // namespace ptk_mad
