/*
    Long arithmetic on k limbs of n bits as circom FUNCTIONS, in the shape of 0xPARC circom-ecdsa's bigint_func.circom
    (circom-ecdsa is not part of the reference tree; the algorithms are the published ones: Knuth D with the
    normalisation 2^n \ (1 + b[k-1]), schoolbook products with deferred carries).  A template calls them with SIGNAL
    values: the comparisons below are run-time branches, the loops over limbs are compile-time loops.
    Arrays carry their exact sizes (circom-ecdsa over-allocates 100 entries everywhere; a register per entry is what a
    run-time function costs here).  The circom text of circom_amd/circuits/bigint.py.
*/
pragma circom 2.0.0;

// 1 when a > b on k limbs
function long_gt(n, k, a, b) {
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

// a - b on k limbs, a >= b
function long_sub(n, k, a, b) {
    var diff[k];
    var borrow[k];
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

// a * b for a single-limb a and a k-limb b: k + 1 limbs
function long_scalar_mult(n, k, a, b) {
    var out[k + 1];
    for (var i = 0; i < k; i++) {
        var temp = out[i] + (a * b[i]);
        out[i] = temp % (1 << n);
        out[i + 1] = out[i + 1] + temp \ (1 << n);
    }
    return out;
}

// quotient digit of a (k + 1 limbs) by b (k limbs, top limb >= 2^(n-1))
function short_div_norm(n, k, a, b) {
    var qhat = (a[k] * (1 << n) + a[k - 1]) \ b[k - 1];
    if (qhat > (1 << n) - 1) {
        qhat = (1 << n) - 1;
    }
    var bext[k + 1];
    for (var i = 0; i < k; i++) {
        bext[i] = b[i];
    }
    var mult[k + 1] = long_scalar_mult(n, k, qhat, b);
    if (long_gt(n, k + 1, mult, a) == 1) {
        mult = long_sub(n, k + 1, mult, bext);
        if (long_gt(n, k + 1, mult, a) == 1) {
            return qhat - 2;
        } else {
            return qhat - 1;
        }
    } else {
        return qhat;
    }
}

// quotient digit of a (k + 1 limbs) by b (k limbs), a < 2^n * b
function short_div(n, k, a, b) {
    var scale = (1 << n) \ (1 + b[k - 1]);
    var norm_a[k + 2] = long_scalar_mult(n, k + 1, scale, a);
    var norm_b[k + 1] = long_scalar_mult(n, k, scale, b);
    return short_div_norm(n, k, norm_a, norm_b);
}

// a (k + m limbs) = out[0] (m + 1 limbs) * b (k limbs) + out[1] (k limbs), b[k - 1] != 0
function long_div(n, k, m, a, b) {
    var w = m + 1 > k ? m + 1 : k;
    var out[2][w];
    var remainder[m + k];
    for (var i = 0; i < m + k; i++) {
        remainder[i] = a[i];
    }
    var dividend[k + 1];
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
        out[0][i] = short_div(n, k, dividend, b);
        var mult_shift[k + 1] = long_scalar_mult(n, k, out[0][i], b);
        var subtrahend[m + k];
        for (var j = 0; j <= k; j++) {
            if (i + j < m + k) {
                subtrahend[i + j] = mult_shift[j];
            }
        }
        remainder = long_sub(n, m + k, remainder, subtrahend);
    }
    for (var i = 0; i < k; i++) {
        out[1][i] = remainder[i];
    }
    return out;
}
