/*
    Big-integer arithmetic on k registers of n bits in the STYLE of 0xPARC circom-ecdsa's bigint.circom / bigint_func.circom
    (written from the published constructions, not a copy: circom-ecdsa is not in the reference tree): unit adders / subtracters
    with explicit carries, BigAdd / BigSub / BigLessThan, BigMult as a carry-less polynomial product whose registers are then
    split with run-time hints, BigMod with the quotient and remainder hinted by a long division FUNCTION over over-allocated
    `var x[100]` arrays (the library's habit: every helper returns 100 entries whatever k is - what exercises the reference's
    "smaller array into a larger variable" rule), BigMultModP on top.
*/
pragma circom 2.0.0;

include "bitify.circom";
include "comparators.circom";

// ---- functions, circom-ecdsa style: fixed 100-entry arrays ------------------------------------------------------------------
function e_long_gt(n, k, a, b) {
    for (var i = k - 1; i >= 0; i--) {
        if (a[i] > b[i]) {
            return 1;
        }
        if (a[i] < b[i]) {
            return 0;
        }
    }
    return 0;
}

function e_long_sub(n, k, a, b) {
    var diff[100];
    var borrow[100];
    for (var i = 0; i < k; i++) {
        if (i == 0) {
            if (a[i] >= b[i]) {
                diff[i] = a[i] - b[i];
                borrow[i] = 0;
            } else {
                diff[i] = a[i] - b[i] + (1 << n);
                borrow[i] = 1;
            }
        } else {
            if (a[i] >= b[i] + borrow[i - 1]) {
                diff[i] = a[i] - b[i] - borrow[i - 1];
                borrow[i] = 0;
            } else {
                diff[i] = (1 << n) + a[i] - b[i] - borrow[i - 1];
                borrow[i] = 1;
            }
        }
    }
    return diff;
}

function e_long_scalar_mult(n, k, a, b) {
    var out[100];
    for (var i = 0; i < 100; i++) {
        out[i] = 0;
    }
    for (var i = 0; i < k; i++) {
        var temp = out[i] + (a * b[i]);
        out[i] = temp % (1 << n);
        out[i + 1] = out[i + 1] + temp \ (1 << n);
    }
    return out;
}

function e_short_div_norm(n, k, a, b) {
    var qhat = (a[k] * (1 << n) + a[k - 1]) \ b[k - 1];
    if (qhat > (1 << n) - 1) {
        qhat = (1 << n) - 1;
    }
    var mult[100] = e_long_scalar_mult(n, k, qhat, b);
    if (e_long_gt(n, k + 1, mult, a) == 1) {
        mult = e_long_sub(n, k + 1, mult, b);
        if (e_long_gt(n, k + 1, mult, a) == 1) {
            return qhat - 2;
        } else {
            return qhat - 1;
        }
    } else {
        return qhat;
    }
}

function e_short_div(n, k, a, b) {
    var scale = (1 << n) \ (1 + b[k - 1]);
    var norm_a[200] = e_long_scalar_mult(n, k + 1, scale, a);      // a 100-entry result into a 200-entry variable
    var norm_b[200] = e_long_scalar_mult(n, k, scale, b);
    var ret;
    if (norm_b[k] != 0) {
        ret = e_short_div_norm(n, k + 1, norm_a, norm_b);
    } else {
        ret = e_short_div_norm(n, k, norm_a, norm_b);
    }
    return ret;
}

// a (k + m registers) = out[0] (m + 1 registers) * b (k registers) + out[1] (k registers)
function e_long_div(n, k, m, a, b) {
    var out[2][100];
    var remainder[200];
    for (var i = 0; i < m + k; i++) {
        remainder[i] = a[i];
    }
    var mult[200];
    var dividend[200];
    for (var i = m; i >= 0; i--) {
        if (i == m) {
            dividend[k] = 0;
            for (var j = k - 1; j >= 0; j--) {
                dividend[j] = remainder[j + m];
            }
        } else {
            for (var j = k; j >= 0; j--) {
                dividend[j] = remainder[j + i];
            }
        }
        out[0][i] = e_short_div(n, k, dividend, b);
        var mult_shift[100] = e_long_scalar_mult(n, k, out[0][i], b);
        var subtrahend[200];
        for (var j = 0; j < m + k; j++) {
            subtrahend[j] = 0;
        }
        for (var j = 0; j <= k; j++) {
            if (i + j < m + k) {
                subtrahend[i + j] = mult_shift[j];
            }
        }
        remainder = e_long_sub(n, m + k, remainder, subtrahend);      // 100 entries into 200
    }
    for (var i = 0; i < k; i++) {
        out[1][i] = remainder[i];
    }
    out[1][k] = 0;
    return out;
}

// ---- unit gadgets ---------------------------------------------------------------------------------------------------------------
template ModSum(n) {
    assert(n <= 252);
    signal input a;
    signal input b;
    signal output sum;
    signal output carry;
    component n2b = Num2Bits(n + 1);
    n2b.in <== a + b;
    carry <== n2b.out[n];
    sum <== a + b - carry * (1 << n);
}

template ModSub(n) {
    assert(n <= 252);
    signal input a;
    signal input b;
    signal output out;
    signal output borrow;
    component lt = LessThan(n);
    lt.in[0] <== a;
    lt.in[1] <== b;
    borrow <== lt.out;
    out <== borrow * (1 << n) + a - b;
}

// a - b - c with the borrow of the whole difference
template ModSubThree(n) {
    assert(n + 2 <= 253);
    signal input a;
    signal input b;
    signal input c;
    signal output out;
    signal output borrow;
    signal b_plus_c;
    b_plus_c <== b + c;
    component lt = LessThan(n + 1);
    lt.in[0] <== a;
    lt.in[1] <== b_plus_c;
    borrow <== lt.out;
    out <== borrow * (1 << n) + a - b_plus_c;
}

template BigAdd(n, k) {
    assert(n <= 252);
    signal input a[k];
    signal input b[k];
    signal output out[k + 1];
    component unit0 = ModSum(n);
    unit0.a <== a[0];
    unit0.b <== b[0];
    out[0] <== unit0.sum;
    component unit[k - 1];
    signal carries[k];
    carries[0] <== unit0.carry;
    for (var i = 1; i < k; i++) {
        unit[i - 1] = ModSum(n);
        unit[i - 1].a <== a[i] + carries[i - 1];       // a[i] + carry < 2^n + 1: ModSum's n + 1 bits hold it
        unit[i - 1].b <== b[i];
        out[i] <== unit[i - 1].sum;
        carries[i] <== unit[i - 1].carry;
    }
    out[k] <== carries[k - 1];
}

// a - b for a >= b; underflow = 1 otherwise
template BigSub(n, k) {
    assert(n <= 252);
    signal input a[k];
    signal input b[k];
    signal output out[k];
    signal output underflow;
    component unit0 = ModSub(n);
    unit0.a <== a[0];
    unit0.b <== b[0];
    out[0] <== unit0.out;
    component unit[k - 1];
    signal borrows[k];
    borrows[0] <== unit0.borrow;
    for (var i = 1; i < k; i++) {
        unit[i - 1] = ModSubThree(n);
        unit[i - 1].a <== a[i];
        unit[i - 1].b <== b[i];
        unit[i - 1].c <== borrows[i - 1];
        out[i] <== unit[i - 1].out;
        borrows[i] <== unit[i - 1].borrow;
    }
    underflow <== borrows[k - 1];
}

template BigLessThan(n, k) {
    signal input a[k];
    signal input b[k];
    signal output out;
    component lt[k];
    component eq[k];
    for (var i = 0; i < k; i++) {
        lt[i] = LessThan(n);
        lt[i].in[0] <== a[i];
        lt[i].in[1] <== b[i];
        eq[i] = IsEqual();
        eq[i].in[0] <== a[i];
        eq[i].in[1] <== b[i];
    }
    // from the top register down: smaller there, or equal there and smaller below
    signal acc[k];
    acc[0] <== lt[0].out;
    for (var i = 1; i < k; i++) {
        acc[i] <== lt[i].out + eq[i].out * acc[i - 1];
    }
    out <== acc[k - 1];
}

// carry-less product: 2k - 1 registers of up to 2n + log2(k) bits, pinned by the polynomial identity at 2k - 1 points
template BigMultNoCarry(n, k) {
    signal input a[k];
    signal input b[k];
    signal output out[2 * k - 1];
    var prod_val[2 * k - 1];
    for (var i = 0; i < 2 * k - 1; i++) {
        prod_val[i] = 0;
        if (i < k) {
            for (var a_idx = 0; a_idx <= i; a_idx++) {
                prod_val[i] = prod_val[i] + a[a_idx] * b[i - a_idx];
            }
        } else {
            for (var a_idx = i - k + 1; a_idx < k; a_idx++) {
                prod_val[i] = prod_val[i] + a[a_idx] * b[i - a_idx];
            }
        }
        out[i] <-- prod_val[i];
    }
    var a_poly[2 * k - 1];
    var b_poly[2 * k - 1];
    var out_poly[2 * k - 1];
    for (var x = 0; x < 2 * k - 1; x++) {
        out_poly[x] = 0;
        a_poly[x] = 0;
        b_poly[x] = 0;
        for (var i = 0; i < 2 * k - 1; i++) {
            out_poly[x] = out_poly[x] + out[i] * (x ** i);
        }
        for (var i = 0; i < k; i++) {
            a_poly[x] = a_poly[x] + a[i] * (x ** i);
            b_poly[x] = b_poly[x] + b[i] * (x ** i);
        }
        out_poly[x] === a_poly[x] * b_poly[x];
    }
}

// proper registers of a carry-less product: split each register into n bits + carry, add the carries along
template LongToShortNoEndCarry(n, k) {
    assert(n <= 126);
    signal input in[k];
    signal output out[k + 1];
    var split[k][3];
    for (var i = 0; i < k; i++) {
        split[i][0] = in[i] % (1 << n);
        split[i][1] = (in[i] \ (1 << n)) % (1 << n);
        split[i][2] = in[i] \ (1 << (2 * n));
    }
    var carry[k];
    carry[0] = 0;
    var running[k + 1];
    running[0] = split[0][0];
    if (k > 1) {
        var sumAndCarry = split[0][1] + split[1][0];
        running[1] = sumAndCarry % (1 << n);
        carry[1] = sumAndCarry \ (1 << n);
    }
    if (k > 2) {
        for (var i = 2; i < k; i++) {
            var sumAndCarry2 = split[i][0] + split[i - 1][1] + split[i - 2][2] + carry[i - 1];
            running[i] = sumAndCarry2 % (1 << n);
            carry[i] = sumAndCarry2 \ (1 << n);
        }
        running[k] = split[k - 1][1] + split[k - 2][2] + carry[k - 1];
    }
    if (k == 2) {
        running[2] = split[1][1] + split[0][2] + carry[1];
    }
    component outRangeChecks[k + 1];
    for (var i = 0; i < k + 1; i++) {
        out[i] <-- running[i];
        outRangeChecks[i] = Num2Bits(n + 2);
        outRangeChecks[i].in <== out[i];
    }
    // the two register strings denote the same number: compare them as field elements (k * n + 2 n + 8 < 253 asked below)
    var lhs = 0;
    var rhs = 0;
    for (var i = 0; i < k; i++) {
        lhs = lhs + in[i] * (1 << (n * i));
    }
    for (var i = 0; i < k + 1; i++) {
        rhs = rhs + out[i] * (1 << (n * i));
    }
    lhs === rhs;
}

template BigMult(n, k) {
    assert(2 * k * n + 8 <= 252);          // the packed comparison of LongToShortNoEndCarry must fit the field
    signal input a[k];
    signal input b[k];
    signal output out[2 * k];
    component mult = BigMultNoCarry(n, k);
    for (var i = 0; i < k; i++) {
        mult.a[i] <== a[i];
        mult.b[i] <== b[i];
    }
    component longshort = LongToShortNoEndCarry(n, 2 * k - 1);
    for (var i = 0; i < 2 * k - 1; i++) {
        longshort.in[i] <== mult.out[i];
    }
    for (var i = 0; i < 2 * k; i++) {
        out[i] <== longshort.out[i];
    }
}

// a (2k registers) = div (k + 1) * b (k) + mod (k), mod < b
template BigMod(n, k) {
    assert(n <= 126);
    signal input a[2 * k];
    signal input b[k];
    signal output div[k + 1];
    signal output mod[k];
    var longdiv[2][100] = e_long_div(n, k, k, a, b);
    for (var i = 0; i < k; i++) {
        div[i] <-- longdiv[0][i];
        mod[i] <-- longdiv[1][i];
    }
    div[k] <-- longdiv[0][k];
    component div_range_checks[k + 1];
    for (var i = 0; i <= k; i++) {
        div_range_checks[i] = Num2Bits(n);
        div_range_checks[i].in <== div[i];
    }
    component mod_range_checks[k];
    for (var i = 0; i < k; i++) {
        mod_range_checks[i] = Num2Bits(n);
        mod_range_checks[i].in <== mod[i];
    }
    component mul = BigMult(n, k + 1);
    for (var i = 0; i < k; i++) {
        mul.a[i] <== div[i];
        mul.b[i] <== b[i];
    }
    mul.a[k] <== div[k];
    mul.b[k] <== 0;
    component add = BigAdd(n, 2 * k + 2);
    for (var i = 0; i < 2 * k; i++) {
        add.a[i] <== mul.out[i];
        if (i < k) {
            add.b[i] <== mod[i];
        } else {
            add.b[i] <== 0;
        }
    }
    add.a[2 * k] <== mul.out[2 * k];
    add.a[2 * k + 1] <== mul.out[2 * k + 1];
    add.b[2 * k] <== 0;
    add.b[2 * k + 1] <== 0;
    for (var i = 0; i < 2 * k; i++) {
        add.out[i] === a[i];
    }
    add.out[2 * k] === 0;
    add.out[2 * k + 1] === 0;
    add.out[2 * k + 2] === 0;
    component lt = BigLessThan(n, k);
    for (var i = 0; i < k; i++) {
        lt.a[i] <== mod[i];
        lt.b[i] <== b[i];
    }
    lt.out === 1;
}

template BigMultModPStyle(n, k) {
    signal input a[k];
    signal input b[k];
    signal input p[k];
    signal output out[k];
    component big_mult = BigMult(n, k);
    for (var i = 0; i < k; i++) {
        big_mult.a[i] <== a[i];
        big_mult.b[i] <== b[i];
    }
    component big_mod = BigMod(n, k);
    for (var i = 0; i < 2 * k; i++) {
        big_mod.a[i] <== big_mult.out[i];
    }
    for (var i = 0; i < k; i++) {
        big_mod.b[i] <== p[i];
    }
    for (var i = 0; i < k; i++) {
        out[i] <== big_mod.mod[i];
    }
}
